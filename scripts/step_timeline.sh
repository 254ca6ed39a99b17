R="$(pwd)"; cd /tmp && export TMPDIR=/tmp; rm -rf "$R/gpurun_out/tl"
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/tl" -o tl -- python3 "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-closed-loop --no-info-leg --no-sweep > /dev/null 2> "$R/gpurun_out/tl.err"
cd "$R"; python3 scripts/step_timeline.py gpurun_out/tl; rm -rf gpurun_out/tl
