import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: refine-list length per iteration of the benchmark pair (TC_DEBUG=64 prints the running total per call: calls of 1 .. 50
iterations, differences = the queries of iteration i)."""
import subprocess, re
code = r'''
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
src, tgt, T = synth.registration_pair(1000000, seed=1, transform=synth.harness_transform(), noise_sigma=1e-4)
dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
th, sh = tc.Cloud(ctx, dt), tc.Cloud(ctx, ds)
th.estimate_normals(16, out=False)
for it in range(1, 51):
    sh.icp_point_to_plane(th, None, it, None, 0.0)
'''
out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TC_DEBUG="64"), capture_output=True, text=True).stderr
tot = [int(m.group(1)) for m in re.finditer(r"refine queries total (\d+)", out)]
print("refine queries per iteration:", [b - a for a, b in zip([0] + tot, tot)])
