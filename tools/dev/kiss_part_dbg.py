import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: why variants_fuzz.kiss_parting says what it says for one case: python tools/dev/kiss_part_dbg.py <seed> <case>"""
import numpy as np, traceback
import threecrate_amd as tc
from oracle import oracle as O
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import variants_fuzz as V
seed, case = int(sys.argv[1]), int(sys.argv[2])
ctx = tc.GpuContext(0)
cs = V.build_case(seed, case, ctx)
src, tgt, init, P = cs["src"], cs["tgt"], cs["init"], cs["params"]
vs, mx, mn, it = P["vs"], P["mx"], P["mn"], P["it"]
d = np.linalg.norm(src.astype(np.float32), axis=1)
sd = O.voxel_grid_filter(np.ascontiguousarray(src[(d >= np.float32(mn)) & (d <= np.float32(mx))]), vs)
td = tgt
r, nd = O.kiss_icp(src, tgt, init, vs, mx, mn, 1)
print("sd", len(sd), "td", len(td), "nd", nd, "pairs", len(r.correspondences), "max tgt idx", int(np.asarray(r.correspondences)[:, 1].max()), "n tgt", len(tgt))
try:
    print("kiss_parting:", V.kiss_parting(cs, ctx))
except Exception:
    traceback.print_exc()
