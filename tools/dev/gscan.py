import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: normals kernel time against the grid's dimensions (TC_NORMALS_CELL_MULT is read once per process: one process per value).
usage: python tools/dev/gscan.py <mult> [k]   (prints: mult, kernel us; run with TC_DEBUG=256 to see the grid line)"""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = tc.GpuContext(0)
d = torch.from_numpy(synth.uniform_cloud(1_000_000, 2)).cuda()
ctx.estimate_normals(d, k)
ctx.profile_enable(1); ctx.profile_reset()
for _ in range(3): ctx.estimate_normals(d, k)
st = ctx.profile_read(); ctx.profile_enable(0)
print(f"RESULT mult {sys.argv[1]} k {k} us {1e3 * st['normals_knn_pca'][1] / st['normals_knn_pca'][0]:.1f}", flush=True)
