"""does tests/test_gpu_parity.py::test_device_inputs_are_ordered_after_the_torch_stream have teeth?  Same scenario with the ordering
switched off: the result must (usually) be wrong."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
n = 400_000
good = torch.from_numpy(synth.uniform_cloud(n, 61)).cuda()
ref = ctx.estimate_normals(good, 12)
torch.cuda.synchronize()
tc.GpuContext._order = lambda self, dev: None
wrong = 0
for trial in range(3):
    stale = torch.full((n, 3), 7.0 + trial, device="cuda"); del stale
    a = torch.randn(4096, 4096, device="cuda")
    for _ in range(60):
        a = (a @ a) * 1e-4
    late = good + 0.0
    got = ctx.estimate_normals(late, 12)
    wrong += int(not torch.equal(got, ref))
    torch.cuda.synchronize()
print("without the ordering:", wrong, "of 3 results wrong")
