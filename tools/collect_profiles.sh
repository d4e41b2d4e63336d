#!/bin/bash
# Run ON THE GPU BOX (through gpurun):  bash tools/collect_profiles.sh r01
# Collects, for `python3 bench.py`, (1) the rocprofv3 kernel-trace --stats summary and
# (2) HBM traffic counters in separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass:
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), into gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-copy-probe --no-extras > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-copy-probe --no-extras > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-copy-probe --no-extras > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-copy-probe --no-extras > $OUT/bench_sq.log 2>&1
# unit counters for the bound claims of the bench line's roofline.table (round 6): issue, L1 look-ups, LDS, occupancy
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-copy-probe --no-extras > $OUT/bench_sq2.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_tcp -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-copy-probe --no-extras > $OUT/bench_tcp.log 2>&1
python3 tools/summarize_profiles.py $OUT > $OUT/summary.txt 2>&1
tail -30 $OUT/summary.txt
