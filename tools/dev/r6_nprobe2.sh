#!/bin/bash
# Run ON THE GPU BOX: round 6 item 1, second call: LDS gather layouts, f32 vs u32 keys, the uniform-read upper bound with counters, lock-step statistics
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/dev/micro/lds_gather.hip -o /tmp/lds_gather && timeout 300 /tmp/lds_gather | tee gpurun_out/r6/lds_gather.txt
bash tools/dev/ab_normals.sh nkeysu32 nnoprune nuniform 2>&1 | tee gpurun_out/r6/ab_normals2.txt
TC_HIP_LIB="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_nstats.so" timeout 300 python3 tools/dev/npmc.py 2>&1 | grep "\[tc\]" | tail -4 | tee gpurun_out/r6/nstats.txt
for v in nnoprune nuniform; do
  export TC_HIP_LIB="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
  i=0; mkdir -p gpurun_out/r6/pmc_$v
  for line in "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" \
              "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
              "TD_TD_BUSY_sum TD_TC_STALL_sum TA_TA_BUSY_sum TA_BUSY_max"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $line --output-format csv -d gpurun_out/r6/pmc_$v/p$i -- python3 tools/dev/npmc.py > gpurun_out/r6/pmc_$v/log$i.txt 2>&1
  done
done
unset TC_HIP_LIB
python3 - <<'PY' | tee gpurun_out/r6/counters2.txt
import glob, csv, collections, os
for d in sorted(glob.glob("gpurun_out/r6/pmc_*")):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void tc::", "").replace("tc::", "")
            a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k in agg:
        if "normals_tagged" not in k and "normals_knn" not in k: continue
        print(os.path.basename(d), k)
        for c, (v, n) in sorted(agg[k].items()): print(f"    {c:42s} {v/n/1e6:12.3f} M/launch  ({n} launches)")
PY
timeout 900 python3 -m pytest tests -m gpu -x -q -k "normals" 2>&1 | tail -5 | tee gpurun_out/r6/pytest_normals.txt
