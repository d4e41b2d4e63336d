import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: two settings of the normals kernel family (TC_NORMALS_TAG = 0 register list, 1 tagged keys, 2 + flattened row groups, 3 / unset
policy) on 26 clouds, bit for bit, + kernel times.
usage: python tools/dev/nfast_ab.py run <tag>      (in a process with TC_NORMALS_TAG set as wanted; writes /tmp/nfast_<tag>.npz)
       python tools/dev/nfast_ab.py cmp <a> <b>"""
import numpy as np


def clouds():
    from threecrate_amd import synth
    rng = np.random.default_rng(3)
    yield "u1M_s1_noisy", synth.registration_pair(1_000_000, seed=1, transform=synth.harness_transform(), noise_sigma=1e-4)[1], (16, 10)
    yield "u1M_s2", synth.uniform_cloud(1_000_000, 2), (16, 8, 20)
    yield "u300k", synth.uniform_cloud(300_000, 5), (16,)
    yield "u10M_slab", synth.uniform_cloud(3_000_000, 7, (10.0, 10.0, 1.0)), (16,)
    yield "tum", synth.tum_shaped_cloud(seed=1), (16, 10)
    yield "kitti", synth.kitti_shaped_cloud(seed=2), (16, 10)
    yield "u20k", synth.uniform_cloud(20000, 1), (16, 10, 5)
    yield "u500", synth.uniform_cloud(500, 1), (16, 3)
    yield "u12", synth.uniform_cloud(12, 1), (16, 10)
    g = np.stack(np.meshgrid(np.arange(30), np.arange(30), np.arange(30), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.1
    yield "lattice", g, (16, 6)
    d = synth.uniform_cloud(50000, 9); d[::7] = d[1::7][: len(d[::7])]
    yield "dups", d, (16,)
    c = (rng.normal(0, 1, (200000, 3)) * np.array([1, 1, 0.02])).astype(np.float32)
    yield "gauss_sheet", c, (16, 10)
    o = synth.uniform_cloud(200000, 4); o[0] = (50, 50, 50)
    yield "outlier", o, (16,)


def run(tag):
    import torch, threecrate_amd as tc
    ctx = tc.GpuContext(0)
    out = {}
    for name, pts, ks in clouds():
        d = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
        for k in ks:
            r = ctx.estimate_normals(d, k)
            ctx.profile_enable(1); ctx.profile_reset()
            for _ in range(3): r = ctx.estimate_normals(d, k)
            st = ctx.profile_read(); ctx.profile_enable(0)
            us = 1e3 * st["normals_knn_pca"][1] / st["normals_knn_pca"][0]
            print(f"{tag} {name} k={k}: normals_knn_pca {us:.1f} us", flush=True)
            out[f"{name}_k{k}"] = r.cpu().numpy()
            if name.startswith("u1M") and k == 16:
                h = tc.Cloud(ctx, d)
                ctx.profile_enable(1); ctx.profile_reset()
                out[f"{name}_k{k}_handle"] = h.estimate_normals(k).cpu().numpy()
                st = ctx.profile_read(); ctx.profile_enable(0)
                print(f"{tag} {name} k={k} handle grid: normals_knn_pca {1e3 * st['normals_knn_pca'][1]:.1f} us", flush=True)
                h.close()
    np.savez(f"/tmp/nfast_{tag}.npz", **out)


def cmp(a, b):
    A, B = np.load(f"/tmp/nfast_{a}.npz"), np.load(f"/tmp/nfast_{b}.npz")
    bad = 0
    for k in A.files:
        same = np.array_equal(A[k].view(np.uint32), B[k].view(np.uint32))
        nd = int((A[k].view(np.uint32) != B[k].view(np.uint32)).any(1).sum())
        print(f"{k}: {'identical' if same else f'{nd} rows differ'}")
        bad += 0 if same else 1
    print("ALL IDENTICAL" if bad == 0 else f"{bad} cases differ")
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "run": run(sys.argv[2])
    else: sys.exit(1 if cmp(sys.argv[2], sys.argv[3]) else 0)
