#!/usr/bin/env python3
"""Summarises a tools/collect_profiles.sh run: per-kernel stats + per-launch HBM traffic.

gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE is reported in KiB and counts
128-B requests as 64 B for wide coalesced streams, i.e. up to 2x low; WRITE_SIZE (KiB) is exact for
16-B-per-lane stores.  Both raw and the x2-corrected read figure are reported; our kernels mix
4/16-B gathers with streams, so the truth lies between them.
"""
import os, re, collections
import csv
import glob
import json
import sys

out = sys.argv[1]
res = {}


def load(sub, pat):
    # (gpurun merges every call's files into the same local directory: take the newest run)
    fs = sorted(glob.glob(f"{out}/{sub}/*/*{pat}.csv"), key=os.path.getmtime)
    return list(csv.DictReader(open(fs[-1]))) if fs else []


def short(name):
    # template arguments = ICP mode (0 point-to-point, 1 point-to-plane, 2 GICP) and, for the main pass, whether the launch also
    # runs the previous iteration's finalize step in its solver block ("fused": DESIGN 4.3)
    # (round 6: <MODE, STATS, CERT> -- the counting instantiation and the second-neighbour certificate's instantiation get a suffix)
    # (a chunk's last main pass also counts its searching lanes: <MODE, false, false, true> -- the same kernel plus one scalar counter; it is
    # folded into the plain row: the bench line's average is over both)
    m = re.search(r"icp_correspond_reduce_kernel<(\d)(?:, ?(true|false|1|0))?(?:, ?(true|false|1|0))?(?:, ?(true|false|1|0))?>", name)
    if m:
        return f"icp_correspond_reduce_kernel<{m.group(1)}>" + (" stats" if m.group(2) in ("true", "1") else "") + (" cert" if m.group(3) in ("true", "1") else "")
    for k in ["icp_correspond_reduce_kernel<1>", "icp_correspond_reduce_kernel<0>", "icp_correspond_reduce_kernel<2>",
              "icp_refine_kernel<1>", "icp_refine_kernel<0>", "icp_refine_kernel<2>", "icp_finalize_kernel", "knn_kernel",
              "bin_count_kernel", "bin_offsets_kernel", "bin_scatter_kernel", "bin_place_kernel", "bbox_state_init_kernel", "vox_hist_kernel", "vox_scatter_kernel",
              "vox_rank_kernel", "vox_flag_kernel", "vox_centroid_kernel",
              "normals_tagged_kernel", "normals_knn_pca_kernel", "normals_coop_kernel", "normals_overflow_kernel", "cell_hist_kernel", "place_kernel", "rerank_kernel", "scatter_kernel",
              "rank_gather_kernel", "scan_apply_kernel", "scan_top_kernel", "scan_reduce_kernel", "bbox_kernel",
              "gather_normals_kernel", "icp_finish_kernel", "icp_write_corr_kernel", "icp_final_mse_kernel"]:
        if k.split("<")[0] in name and (("<" not in k) or (k[k.index("<"):] in name)):
            return k
    return name[:60]


stats = load("stats", "kernel_stats")
print("== rocprofv3 --kernel-trace --stats (python3 bench.py --steps 3 --warmup 1) ==")
print(f"{'kernel':46s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
# (instantiations that share a short name -- the plain main pass and its counting twin at the end of every chunk -- are ONE row: calls
# and total time added, the average over both)
merged = collections.OrderedDict()
for r in stats:
    m = merged.setdefault(short(r["Name"]), {"calls": 0, "total_ns": 0.0, "pct": 0.0})
    m["calls"] += int(r["Calls"]); m["total_ns"] += float(r["TotalDurationNs"]); m["pct"] += float(r["Percentage"])
for k, m in sorted(merged.items(), key=lambda kv: -kv[1]["total_ns"]):
    avg = m["total_ns"] / max(m["calls"], 1) / 1e3
    print(f"{k:46s} {m['calls']:>6d} {avg:10.2f} {m['total_ns']/1e6:10.3f} {m['pct']:6.2f}")
    res.setdefault(k, {})["avg_us"] = avg
    res[k]["calls"] = m["calls"]

for sub, cname in [("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")]:
    acc = collections.defaultdict(list)
    for r in load(sub, "counter_collection"):
        if r["Counter_Name"] == cname:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res.setdefault(k, {})[cname + "_KiB_per_launch"] = sum(v) / len(v)

sq = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_sq", "pmc_sq2", "pmc_tcp"):
    for r in load(sub, "counter_collection"):
        sq[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sq.items():
    for c, v in d.items():
        res.setdefault(k, {})[c] = sum(v) / len(v)

print("\n== HBM traffic per launch (separate --pmc passes) ==")
print(f"{'kernel':46s} {'fetch_MB':>10s} {'fetch_x2_MB':>12s} {'write_MB':>10s}")
for k, d in res.items():
    if "FETCH_SIZE_KiB_per_launch" in d or "WRITE_SIZE_KiB_per_launch" in d:
        f = d.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024 / 1e6
        w = d.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024 / 1e6
        d["traffic_bytes_raw"] = (f + w) * 1e6
        d["traffic_bytes_corrected"] = (2 * f + w) * 1e6
        print(f"{k:46s} {f:10.2f} {2*f:12.2f} {w:10.2f}")
print("\n== SQ counters per launch (millions) ==")
for k, d in res.items():
    if "SQ_INSTS_VALU" in d:
        print(f"{k:46s} " + " ".join(f"{c.replace('SQ_','')}={d[c]/1e6:.2f}" for c in sorted(d) if c.startswith("SQ_")))
# derived per launch (round 6, SURVEY 8d secondary figures): achieved waves per SIMD = wave-cycles / busy CU-cycles (SQ_WAVE_CYCLES counts
# quad-cycles per wave, a CU has four SIMDs: the factors cancel), LDS bank-conflict rate = conflict cycles / LDS-array cycles
print("\n== unit counters per launch ==")
for k, d in res.items():
    if "SQ_BUSY_CU_CYCLES" in d and d["SQ_BUSY_CU_CYCLES"] > 0:
        if "SQ_WAVE_CYCLES" in d: d["achieved_waves_per_simd"] = d["SQ_WAVE_CYCLES"] / d["SQ_BUSY_CU_CYCLES"]
        if d.get("SQ_LDS_IDX_ACTIVE", 0) > 0: d["lds_bank_conflict_rate"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        print(f"{k:46s} busy_cu_cycles={d['SQ_BUSY_CU_CYCLES']/1e6:.2f}M valu={d.get('SQ_INSTS_VALU', d.get('SQ_ACTIVE_INST_VALU', 0))/1e6:.2f}M "
              f"tcp_lookups={d.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0)/1e6:.2f}M waves/SIMD={d.get('achieved_waves_per_simd', float('nan')):.2f} "
              f"lds_conflict_rate={d.get('lds_bank_conflict_rate', float('nan')):.3f}")
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
