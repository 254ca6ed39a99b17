#!/bin/bash
# bench.py on every BASELINE.json config that fits one GPU (controller-only value + closed loop), one JSON line each;
# configs[0] on its own task (`hovering`), the others on tracking_zigzag.  Redirect the printed table into
# gpurun_out/profiles_out/<round>_all_configs.log to have it committed under profiles/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py --controller mppi --task hovering --N 1024 --no-cpu-baseline > gpurun_out/c0.json 2>/dev/null
python bench.py --controller covo-offline --N 8192 --no-cpu-baseline > gpurun_out/c1.json 2>/dev/null
python bench.py --controller covo-online --N 8192 --no-cpu-baseline > gpurun_out/c2.json 2>/dev/null
python bench.py --controller covo-online --N 65536 --no-cpu-baseline > gpurun_out/c3.json 2>/dev/null
python bench.py --controller covo-offline --N 65536 --no-cpu-baseline > gpurun_out/c3_offline.json 2>/dev/null
python bench.py --controller mppi --N 65536 --no-cpu-baseline > gpurun_out/c3_mppi.json 2>/dev/null
for f in c0 c1 c2 c3 c3_offline c3_mppi; do python - "$f" <<'PY'
import json, sys
f = sys.argv[1]
d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
cl = d.get("closed_loop", {})
print(f, d["config"]["controller"], d["config"].get("task"), d["config"]["N_global"], f'{d["value"]:.0f} steps/s', f'{d["ms_per_step"] * 1e3:.1f} us/step', "closed loop:", {k: ((round(v) if abs(v) >= 100 else round(v, 4)) if isinstance(v, float) else v) for k, v in cl.items()} if isinstance(cl, dict) else cl)
PY
done
