#!/bin/bash
# L2 counters of the Sigma chain's launches at batch 8 (one matrix per XCD) and 32 (four): does a persistent launch's sc1 traffic
# hit the XCD's L2?  (GPU box, from the repo root; counters only, the program itself after `--`)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for b in 8 32; do
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_READ_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum"; do
    rm -rf "$R/gpurun_out/pm"
    rocprofv3 --pmc $set --output-format csv -d "$R/gpurun_out/pm" -o s -- python3 "$R/scripts/sigma_batch_scaling.py" $b > /dev/null 2> "$R/gpurun_out/pm.err"
    f=$(find "$R/gpurun_out/pm" -name "*counter_collection.csv" | head -1)
    echo "batch $b: $set  ($f)"
    python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("   no counter file:", e); sys.exit(0)
for r in rows:
    n = r["Kernel_Name"]
    if "ns_" in n: acc[n.split("(")[0][:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("   ", k.ljust(36), {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
  done
done
rm -rf "$R/gpurun_out/pm"
