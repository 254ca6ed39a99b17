#!/usr/bin/env python
"""Per-kernel means of the two separate rocprofv3 --pmc passes over bench.py (FETCH_SIZE, WRITE_SIZE; KB per
dispatch) -> profiles/<round>_bench_pmc_{fetch,write}_size.csv and profiles/<round>_bench_pmc_summary.json.
FETCH_SIZE is doubled for the rollout's 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM section).
usage: pmc_summary.py [gpurun_out dir] [round tag, default r02]"""
import csv, collections, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out")
RND = sys.argv[2] if len(sys.argv) > 2 else "r02"
out = {}
for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(os.path.join(src, f"pmc_{which}", "bench_counter_collection.csv"))))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(root, "profiles", f"{RND}_bench_pmc_{which}_size.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatches", f"{counter}_KB_mean", f"{counter}_KB_min", f"{counter}_KB_max"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), f"{sum(v) / len(v):.3f}", f"{min(v):.3f}", f"{max(v):.3f}"])
    for k, v in agg.items():
        if k.startswith("void rollout_pipe3_kernel") and "GROUPS" not in k:  # the product rollout (one variant per run)
            out[counter + "_KB"] = sum(v) / len(v)
out["traffic_bytes_per_launch"] = int(round((2.0 * out["FETCH_SIZE_KB"] + out["WRITE_SIZE_KB"]) * 1024))
out["kernel"] = "rollout_pipe3_kernel"
try:
    out["commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"]).decode().strip()
except Exception:
    out["commit"] = None
out["note"] = "rollout_pipe3_kernel at N_local=65536: 2 x FETCH_SIZE (gfx950 16-B/lane correction) + WRITE_SIZE; fabric counters include Infinity-Cache hits"
out["command"] = "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-closed-loop"
json.dump(out, open(os.path.join(root, "profiles", f"{RND}_bench_pmc_summary.json"), "w"), indent=1)
print(out)
