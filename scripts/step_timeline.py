#!/usr/bin/env python
"""Per-step timeline of the covo-online control step from a rocprofv3 kernel trace of bench.py: for the timed steps, every
launch's start offset, duration and the gap to the launch before it (medians over the steps).
usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o tl -- python3 $REPO/bench.py --steps 40 --warmup 5 \\
                  --no-cpu-baseline --no-closed-loop --no-info-leg --no-sweep ;  python3 scripts/step_timeline.py OUT"""
import csv
import glob
import sys
from collections import defaultdict

import numpy as np

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step = from one adj_jac launch to the next
steps, cur = [], []
for r in rows:
    name = r["Kernel_Name"]
    if "adj_jac_kernel" in name and cur:
        steps.append(cur)
        cur = []
    if cur or "adj_jac_kernel" in name:
        cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:44]


steps = [s for s in steps if 5 <= len(s) <= 10 and any("rollout_pipe3" in n for n, _, _ in s) and any("merge_kernel" in n for n, _, _ in s)]
from collections import Counter
shape = Counter(tuple(short(n) for n, _, _ in s) for s in steps).most_common(1)[0][0]
sel = [s for s in steps if tuple(short(n) for n, _, _ in s) == shape][-60:]
print(f"{len(sel)} steps of {len(shape)} launches")
tot = []
for i, nm in enumerate(shape):
    st = np.array([s[i][1] - s[0][1] for s in sel]) / 1e3
    du = np.array([s[i][2] - s[i][1] for s in sel]) / 1e3
    gap = np.array([s[i][1] - s[i - 1][2] for s in sel]) / 1e3 if i else np.zeros(len(sel))
    print(f"{nm:46s} start {np.median(st):7.2f}  dur {np.median(du):6.2f}  gap before {np.median(gap):5.2f} us")
span = np.array([s[-1][2] - s[0][1] for s in sel]) / 1e3
nxt = np.array([sel[i + 1][0][1] - sel[i][-1][2] for i in range(len(sel) - 1)]) / 1e3
print(f"first launch's start -> last launch's end: {np.median(span):.2f} us; last end -> next step's first start: {np.median(nxt):.2f} us")
