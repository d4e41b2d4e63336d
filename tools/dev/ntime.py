import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: median wall time of estimate_normals(k=16) on the 1 M-point bench cloud, device-resident"""
import time, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
d = torch.from_numpy(synth.uniform_cloud(1000000, 2)).cuda()
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.estimate_normals(d, 16); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("normals 1M k=16: median %.1f us  min %.1f us" % (1e6 * np.median(ts[5:]), 1e6 * min(ts[5:])))
