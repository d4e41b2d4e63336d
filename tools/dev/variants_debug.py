import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: take one variants_fuzz.py case apart -- is the GPU result deterministic, does the voxel filter agree with the oracle's
on the case's clouds, how do the plain point-to-point loops on the down-sampled clouds compare.
usage: python tools/dev/variants_debug.py <seed> <case>..."""
import numpy as np
import threecrate_amd as tc
from oracle import oracle as O
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import variants_fuzz as V


def keyed(p):
    return p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]


def main():
    seed = int(sys.argv[1])
    ctx = tc.GpuContext(0)
    for case in map(int, sys.argv[2:]):
        cs = V.build_case(seed, case, ctx)
        print("==", cs["tag"])
        src, tgt, init, which, P = cs["src"], cs["tgt"], cs["init"], cs["which"], cs["params"]
        r = cs["orun"](src)
        print("   oracle:", r.converged, r.iterations, "mse %.9g" % r.mse, np.asarray(r.transformation))
        for rep in range(3):
            g = cs["grun"](src)
            print("   gpu   :", g.converged, g.iterations, "mse %.9g" % g.mse, np.asarray(g.transformation), "frob %.3e" % V.frob(g.transformation, r.transformation))
        voxels = []
        if which == 1: voxels = [P["vs"]]
        if which == 0: voxels = [l[0] for l in P["levels"]]
        for vs in voxels:
            for name, cloud in (("src", src), ("tgt", tgt)):
                if which == 1:
                    d = np.linalg.norm(cloud.astype(np.float32), axis=1)
                    keep = (d >= np.float32(P["mn"])) & (d <= np.float32(P["mx"]))
                    if name == "tgt": keep[:] = True
                    cloud = np.ascontiguousarray(cloud[keep])
                go = np.asarray(ctx.voxel_grid_filter(cloud, vs))
                oo = O.voxel_grid_filter(cloud, vs)
                same = len(go) == len(oo) and np.array_equal(keyed(go), keyed(oo))
                print(f"   voxel {vs:.6g} on {name} ({len(cloud)} pts): gpu {len(go)} oracle {len(oo)} voxels, identical sets: {same}")
                if not same and len(go) == len(oo):
                    dd = np.abs(keyed(go) - keyed(oo)).max()
                    print(f"      max abs difference of the sorted centroids {dd:.3e}")
                # the plain loop on the down-sampled pair
        if which == 0:
            s0, t0 = src, tgt
            for (vs, it, md) in P["levels"]:
                sd, td = O.voxel_grid_filter(s0, vs), O.voxel_grid_filter(t0, vs)
                ro = O.icp_detailed(sd, td, init, it, md, P["thr"])
                rg = ctx.icp_detailed(sd, td, init, it, md, P["thr"], correspondences=False)
                print(f"   level voxel {vs:.4g} iters {it} md {md}: {len(sd)}/{len(td)} pts  oracle {ro.converged} {ro.iterations} mse {ro.mse:.9g} | gpu {rg.converged} {rg.iterations} mse {rg.mse:.9g} frob {V.frob(rg.transformation, ro.transformation):.3e}")


main()
