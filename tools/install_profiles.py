#!/usr/bin/env python3
"""Copies a tools/collect_profiles.sh run (gpurun_out/prof_<tag>/) + the bench lines of the same build into profiles/ under
<name>_* and refreshes profiles/pmc_traffic.json (read by bench.py for roofline.traffic).
usage: python tools/install_profiles.py <tag> <name> [bench.json [aux.json ...]]"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profiles.py"), src], capture_output=True, text=True).stdout
open(os.path.join(dst, f"{name}_summary.txt"), "w").write(out)
shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, f"{name}_summary.json"))
ks = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
if ks: shutil.copy(ks[-1], os.path.join(dst, f"{name}_kernel_stats.csv"))
summ = json.load(open(os.path.join(src, "summary.json")))
UNIT = ("SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "TCP_TOTAL_CACHE_ACCESSES_sum",
        "TCP_PENDING_STALL_CYCLES_sum", "achieved_waves_per_simd", "lds_bank_conflict_rate", "avg_us")
kernels = {k: {kk: vv for kk, vv in v.items() if "SIZE" in kk or "traffic" in kk or kk in UNIT} for k, v in summ.items() if "traffic_bytes_raw" in v}
json.dump({"source": f"profiles/{name}_summary.json (tools/collect_profiles.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                     "`python3 bench.py --steps 2 --warmup 1 --no-extras`)",
           "note": "FETCH_SIZE / WRITE_SIZE in KiB per launch averaged over the launches of the run; traffic_bytes_corrected doubles the fetch "
                   "figure (MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests as 64 B on gfx950 for wide streams), raw does not; the truth "
                   "for these gather-heavy kernels lies between",
           "kernels": kernels}, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
if len(sys.argv) > 3:
    shutil.copy(sys.argv[3], os.path.join(dst, f"{name}_bench.json"))
if len(sys.argv) > 4:
    with open(os.path.join(dst, f"{name}_aux_bench.jsonl"), "w") as f:
        for p in sys.argv[4:]:
            lines = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
            if lines: f.write(lines[-1] + "\n")
print(out[:1500])
