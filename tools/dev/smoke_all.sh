#!/bin/bash
# Run ON THE GPU BOX: does every script of tools/dev/ still start and run against the current library?  
# Each script gets 40 s; exit code 0 or a timeout (the fuzzers run until told to stop) count as "runs".
cd "$GRAFT_REPO_ROOT"
declare -A ARGS=( [loop_case.py]="1 1" [ncase.py]="1 1" [normals_case.py]="1 1 0" [variants_case.py]="1 1" [gscan.py]="1.0" [nfast_ab.py]="run smoke"
                  [farq.py]="run smoke" [nstamps_cloud.py]="uniform" [fuzz.py]="5 1" [loop_fuzz.py]="5 1" [normals_fuzz.py]="5 1" [variants_fuzz.py]="5 1"
                  [vor_fuzz.py]="5 1" [timeline.py]="gpurun_out/step_trace 5" [case.py]="" [dbg.py]="0" )
for f in tools/dev/*.py; do
  b=$(basename $f)
  timeout 40 python3 $f ${ARGS[$b]:-} > /tmp/smoke_$b.log 2>&1
  rc=$?
  st=FAIL; [ $rc -eq 0 ] && st=ok; [ $rc -eq 124 ] && st=ok-timeout
  echo "$st rc=$rc $b $(tail -1 /tmp/smoke_$b.log | cut -c1-110)"
done
