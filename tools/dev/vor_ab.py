"""A/B of the inscribed-ball fast path (TC_DEBUG & 4 disables it): same bits, different time."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, time, json, numpy as np, torch
sys.path.insert(0, %r)
import threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
out = {}
for name, kw in (("uniform", dict(transform=synth.harness_transform(), noise_sigma=1e-4)), ("tsmall", dict())):
    src, tgt, T = synth.registration_pair(1_000_000, seed=1, **kw)
    ds, dt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    nrm = ctx.estimate_normals(dt, 16)
    for mode in ("p2plane", "p2p"):
        f = (lambda: ctx.icp_point_to_plane_detailed(ds, dt, nrm, None, 50, None, 0.0)) if mode == "p2plane" else (lambda: ctx.icp_detailed(ds, dt, None, 50, None, 0.0))
        r = f()
        t0 = time.perf_counter()
        for _ in range(3):
            r = f()
        dt_ms = (time.perf_counter() - t0) / 3 * 1e3
        out[name + "_" + mode] = dict(ms=dt_ms, T=[float(v) for v in r.transformation], mse=r.mse, corr=int(np.asarray(r.correspondences)[:, 1].astype(np.int64).sum()))
print(json.dumps(out))
''' % ROOT
res = {}
for dbg in ("0", "4"):
    env = dict(os.environ, TC_DEBUG=dbg); env.setdefault("TC_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "threecrate_amd", "variants", "libthreecrate_hip_dev.so"))    # the altering bits exist in the dev build only
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(p.stderr[-2000:]); sys.exit(1)
    res[dbg] = json.loads(line[0])
for k in res["0"]:
    a, b = res["0"][k], res["4"][k]
    same = a["T"] == b["T"] and a["mse"] == b["mse"] and a["corr"] == b["corr"]
    print(f"{k:18s} fast path {a['ms']:8.3f} ms   without {b['ms']:8.3f} ms   identical results: {same}")
