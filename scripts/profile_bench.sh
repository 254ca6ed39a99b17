#!/bin/bash
# Everything profiles/ holds for one round, in one GPU call: the unprofiled bench line, the rocprofv3 kernel stats of the
# same command, the two separate --pmc passes (FETCH_SIZE / WRITE_SIZE; counters are never combined with traces), the
# phase times and the stand-alone kernel table.  Run on the GPU box from the repo root; outputs land in gpurun_out/.
R="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" > "$R/gpurun_out/bench_line.json" 2> "$R/gpurun_out/bench_line.err"
rm -rf "$R/gpurun_out/prof_stats" "$R/gpurun_out/pmc_fetch" "$R/gpurun_out/pmc_write"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o bench -- python3 "$R/bench.py" \
    > "$R/gpurun_out/bench_line_under_rocprof.json" 2> "$R/gpurun_out/prof_stats.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch" -o bench -- python3 "$R/bench.py" --steps 30 --warmup 5 \
    --no-cpu-baseline --no-closed-loop > /dev/null 2> "$R/gpurun_out/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write" -o bench -- python3 "$R/bench.py" --steps 30 --warmup 5 \
    --no-cpu-baseline --no-closed-loop > /dev/null 2> "$R/gpurun_out/pmc_write.err"
cd "$R"
python3 scripts/phase_times.py > gpurun_out/phase_times.log 2>/dev/null
python3 scripts/kbench.py > gpurun_out/kbench.log 2>/dev/null
python3 scripts/pmc_summary.py gpurun_out r02 > gpurun_out/pmc_summary.log 2>&1
find gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" | head -20
tail -c 600 gpurun_out/bench_line.json
