#!/bin/bash
# Same-box A/B of two builds of the library: scripts/ab_libs.sh "<bench.py arguments>" <lib A> <lib B> [rounds, default 3]
# (lib = path of a libcovo_hip*.so, e.g. one built from another commit with `make VARIANT=old`).  Prints value and us per step.
ARGS="$1"; A="$2"; B="$3"; N="${4:-3}"
for i in $(seq "$N"); do for L in "$A" "$B"; do
    COVO_HIP_LIB="$L" python bench.py $ARGS 2>/dev/null | tail -1 | L="$L" python -c "
import json, os, sys
d = json.loads(sys.stdin.read())
cl = (d.get('closed_loop') or {}).get('device_env')
print(os.path.basename(os.environ['L']), round(d['value'], 1), round(d['ms_per_step'] * 1e3, 2), 'closed loop', None if cl is None else round(cl, 1))"
done; done
