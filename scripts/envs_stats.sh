#!/bin/bash
# config [4] quick look: the bench line's phases and the rocprofv3 kernel stats of the same command (GPU box, from the repo root)
R="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$R/gpurun_out"
python3 "$R/bench.py" --config envs --steps 100 2>/dev/null > "$R/gpurun_out/envs_line.json"
python3 - "$R/gpurun_out/envs_line.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "closed_loop", (d.get("closed_loop") or {}).get("value"))
for k, v in d.items():
    if isinstance(v, dict) and any(x.endswith("_us") for x in v): print(k, v)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_envs"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_envs" -o envs -- python3 "$R/bench.py" --config envs --steps 30 --warmup 5 --no-closed-loop > /dev/null 2>&1
cd "$R"
find gpurun_out/prof_envs -name "*kernel_stats.csv" -exec cp {} gpurun_out/envs_kernel_stats.csv \;
rm -rf gpurun_out/prof_envs
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/envs_kernel_stats.csv")))
n = max(int(r["Calls"]) for r in rows if "noise_gemm" in r["Name"])
for r in rows[:14]:
    print(r["Name"][:44].ljust(46), r["Calls"].rjust(6), "%8.2f us" % (float(r["AverageNs"]) / 1e3), "%7.1f us/step" % (float(r["TotalDurationNs"]) / 1e3 / n), r["Percentage"])
PY
