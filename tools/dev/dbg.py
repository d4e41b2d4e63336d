import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sys, time, os, subprocess
for dbg in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 16, 2|4, 2|4|8]:
    env = dict(os.environ, TC_DEBUG=str(dbg)); env.setdefault("TC_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "threecrate_amd", "variants", "libthreecrate_hip_dev.so"))    # the altering bits exist in the dev build only
    out = subprocess.run([sys.executable, "tools/dev/gpu.py", "1000000"], env=env, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if "icp_correspond_reduce" in l][-1]
    print(dbg, line.strip())
