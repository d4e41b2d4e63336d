cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/step_trace; mkdir -p gpurun_out/step_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/step_trace/t -- python3 bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline --no-copy-probe > gpurun_out/step_trace/log.txt 2>&1
python3 tools/dev/timeline.py gpurun_out/step_trace 182 --gaps 3.0 | tail -60
