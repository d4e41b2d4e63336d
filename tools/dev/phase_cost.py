"""main-pass cost in the converged state under timing knobs (TC_DEBUG bits 4: no inscribed-ball test, 16: no accumulate phase, 32: transform frozen)"""
import os, subprocess, sys, json
code = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, "/root/repo")
import threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
src, tgt, T = synth.registration_pair(1_000_000, seed=1, transform=synth.harness_transform(), noise_sigma=1e-4)
ds, dt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
nrm = ctx.estimate_normals(dt, 16)
out = {}
for name, init in (("moving", synth.yaw_isometry((0.03, -0.012, 0.006), 0.012)), ("converged", synth.harness_transform())):
    ctx.icp_point_to_plane_detailed(ds, dt, nrm, init, 24, None, 0.0, correspondences=False)
    ctx.profile_enable(1); ctx.profile_reset()
    ctx.icp_point_to_plane_detailed(ds, dt, nrm, init, 24, None, 0.0, correspondences=False)
    st = ctx.profile_read(); ctx.profile_enable(0)
    out[name] = {k: round(1e3 * ms / max(c, 1), 1) for k, (c, ms) in st.items() if k.startswith("icp_")}
print(json.dumps(out))
'''
for label, dbg, extra in (("default", 0, {}), ("vor from it 0", 0, {"TC_VOR_AFTER": "0"}), ("no vor", 4, {}),
                          # no phase A = no sums = the solve fails: freeze the transform instead (every pass then is COLD: no warm start)
                          ("cold", 32, {}), ("cold, no phase A", 32 + 16, {})):
    env = dict(os.environ, TC_DEBUG=str(dbg), **extra); env.setdefault("TC_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "threecrate_amd", "variants", "libthreecrate_hip_dev.so"))    # the altering bits exist in the dev build only
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print(f"{label:22s}", line[0] if line else p.stderr[-1500:])
