import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: where the host-buffer (drop-in) path spends its time: pageable / pinned copy rates, page-locking cost, host memcpy rate,
and the host-path calls with dense and with materialised correspondences"""
import time, ctypes as C, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth, _lib

def med(f, n=7):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2]

n = 1_000_000
src, tgt, T = synth.registration_pair(n, seed=1, transform=synth.harness_transform(), noise_sigma=1e-4)
a = np.random.rand(n, 6).astype(np.float32)            # 24 MB pageable
d = torch.empty(n, 6, device="cuda")
p = torch.empty(n, 6).pin_memory()
t = med(lambda: d.copy_(torch.from_numpy(a))); print(f"H2D pageable 24 MB: {t*1e3:.2f} ms = {24e-3/t:.1f} GB/s")
t = med(lambda: d.copy_(p, non_blocking=True)); print(f"H2D pinned   24 MB: {t*1e3:.2f} ms = {24e-3/t:.1f} GB/s")
t = med(lambda: p.copy_(d, non_blocking=True)); print(f"D2H pinned   24 MB: {t*1e3:.2f} ms = {24e-3/t:.1f} GB/s")
h = torch.from_numpy(a)
t = med(lambda: h.copy_(d)); print(f"D2H pageable 24 MB: {t*1e3:.2f} ms = {24e-3/t:.1f} GB/s")
b = np.empty_like(a)
t = med(lambda: np.copyto(b, a)); print(f"host memcpy 24 MB (1 thread): {t*1e3:.2f} ms = {24e-3/t:.1f} GB/s")
rt = torch.cuda.cudart()
def reg():
    rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); rt.cudaHostUnregister(a.ctypes.data)
try:
    t = med(reg); print(f"hipHostRegister + Unregister 24 MB: {t*1e3:.2f} ms")
except Exception as e:
    print("host register failed", e)
ctx = tc.GpuContext(0)
nrm = ctx.estimate_normals(tgt, 16)
t = med(lambda: ctx.estimate_normals(tgt, 16), 5); print(f"host normals call: {t*1e3:.2f} ms")
t = med(lambda: ctx.icp_point_to_plane_detailed(src, tgt, nrm, None, 50, None, 0.0, correspondences=True), 5); print(f"host ICP call, pairs materialised in Python: {t*1e3:.2f} ms")
t = med(lambda: ctx.icp_point_to_plane_detailed(src, tgt, nrm, None, 50, None, 0.0, correspondences="device"), 5); print(f"host ICP call, dense corr_target array: {t*1e3:.2f} ms")
t = med(lambda: ctx.icp_point_to_plane_detailed(src, tgt, np.ascontiguousarray(nrm[:, 3:]), None, 50, None, 0.0, correspondences="device"), 5); print(f"host ICP call, dense corr, n x 3 normals: {t*1e3:.2f} ms")
t = med(lambda: ctx.icp_point_to_plane_detailed(src, tgt, nrm, None, 50, None, 0.0, correspondences=False), 5); print(f"host ICP call, no correspondences: {t*1e3:.2f} ms")
ds, dt, dn = (torch.from_numpy(x).cuda() for x in (src, tgt, nrm))
t = med(lambda: ctx.icp_point_to_plane_detailed(ds, dt, dn, None, 50, None, 0.0, correspondences="device"), 5); print(f"device ICP call: {t*1e3:.2f} ms")
t = med(lambda: ctx.estimate_normals(dt, 16), 5); print(f"device normals call: {t*1e3:.2f} ms")
