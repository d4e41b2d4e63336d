import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""(TC_DEBUG bits that change the road -- 32, 8192 ... -- need TC_HIP_LIB=threecrate_amd/variants/libthreecrate_hip_dev.so since round 6.)
dev: far queries -- outliers at 30 / 100 extents, partially overlapping scans -- timing + results saved for an A/B of the
refine pass's ball scan (default) against shells only (TC_DEBUG=8192).  usage: farq.py <tag>; farq.py cmp <a> <b>"""
import time, numpy as np


def main(tag):
    import torch, threecrate_amd as tc
    from threecrate_amd import synth
    ctx = tc.GpuContext(0)
    out = {}
    def med(fn, reps=3):
        fn(); ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return 1e3 * float(np.median(ts)), r
    for n in (200000, 1000000):
        base = synth.uniform_cloud(n, 1)
        for name, extra in (("clean", np.zeros((0, 3), np.float32)), ("1 outlier x30", np.array([[30, 0.5, 0.5]], np.float32)),
                            ("3 outliers x100", np.array([[100, 0.5, 0.5], [0.5, -100, 0.2], [0.3, 0.3, 100]], np.float32))):
            pts = np.concatenate([base, extra]).astype(np.float32)
            d = torch.from_numpy(pts).cuda()
            src = torch.from_numpy(synth.apply_isometry(synth.yaw_isometry((-0.05, 0.02, -0.01), -0.02), pts)).cuda()
            nrm = ctx.estimate_normals(d, 16)
            ti, r = med(lambda: ctx.icp_point_to_plane_detailed(src, d, nrm, None, 10, None, 0.0, correspondences="device"))
            print(f"{tag} n={n:8d} {name:16s} icp 10 it {ti:9.2f} ms", flush=True)
            out[f"{n}_{name}_T"] = r.transformation; out[f"{n}_{name}_c"] = r.corr_target.cpu().numpy()
        # partially overlapping scans, no maximum distance (the drop-in default): the source shifted by a fraction of the extent
        d = torch.from_numpy(base).cuda()
        nrm = ctx.estimate_normals(d, 16)
        for frac in (0.0, 0.25, 0.5):
            T = synth.yaw_isometry((-0.05 + frac, 0.02, -0.01), -0.02)
            src = torch.from_numpy(synth.apply_isometry(T, base)).cuda()
            for it in (1, 5):
                ti, r = med(lambda: ctx.icp_point_to_plane_detailed(src, d, nrm, None, it, None, 0.0, correspondences="device"))
                print(f"{tag} n={n:8d} overlap {1 - frac:.2f}  icp {it} it {ti:9.2f} ms", flush=True)
            out[f"{n}_ov{frac}_T"] = r.transformation; out[f"{n}_ov{frac}_c"] = r.corr_target.cpu().numpy()
    np.savez(f"/tmp/farq_{tag}.npz", **out)


if __name__ == "__main__":
    if sys.argv[1] == "cmp":
        A, B = np.load(f"/tmp/farq_{sys.argv[2]}.npz"), np.load(f"/tmp/farq_{sys.argv[3]}.npz")
        bad = 0
        for k in A.files:
            same = np.array_equal(A[k], B[k])
            if not same:
                bad += 1
                print(k, "DIFFERS", (A[k] != B[k]).sum())
        print("ALL IDENTICAL" if bad == 0 else f"{bad} differ")
    else:
        main(sys.argv[1])
