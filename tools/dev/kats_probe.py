import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faulthandler; faulthandler.dump_traceback_later(100, exit=True)
import time
import torch  # noqa
import threecrate_amd as tc
from tests import kats
from tests.backends import GpuBackend
ctx = tc.GpuContext(0)
b = GpuBackend(ctx)
for k in kats.GICP_KATS + kats.KISS_KATS:
    t0 = time.time()
    try:
        k(b); r = "ok"
    except AssertionError as e:
        r = "ASSERT " + str(e)[:100]
    print(f"{k.__name__:28s} {r}  {time.time()-t0:.2f}s", flush=True)
print("closing", flush=True)
ctx.close()
print("closed", flush=True)
