#!/bin/bash
R="$(pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_envs_r06"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_envs_r06" -o envs -- python3 "$R/bench.py" --config envs --steps 50 --warmup 5 --no-closed-loop > /dev/null 2> "$R/gpurun_out/prof_envs_r06.err"
cd "$R"
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_envs_r06/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  pct {r['Percentage']}")
PY
