import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
d = torch.from_numpy(synth.uniform_cloud(1000000, 1)).cuda()
for _ in range(3): ctx.estimate_normals(d, 16)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ctx.estimate_normals(d, 16)
torch.cuda.synchronize(); print(f"normals call {1e2*(time.perf_counter()-t0):.3f} ms")
