#!/usr/bin/env python
"""Per-kernel means of the rocprofv3 --pmc passes over `bench.py --config envs` (scripts/profile_bench.sh) ->
profiles/<round>_bench_envs_pmc_summary.json: the batched rollout launch's HBM traffic (2 x FETCH_SIZE + WRITE_SIZE: the gfx950
16-B/lane correction of MI355X_MICROARCH.md, as for the single step) -- what bench.py --config envs attaches as roofline.traffic --
and the Sigma chain's two batched persistent launches (L2 hit / miss, wait shares, matrix-pipe busy cycles).

usage: pmc_summary_envs.py [gpurun_out dir] [round tag, default r05]"""
import collections, csv, glob, json, os, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out")
RND = sys.argv[2] if len(sys.argv) > 2 else "r06"
sys.path.insert(0, root)
from bench import kernel_src_sha  # noqa: E402


def load(which):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(os.path.join(src, f"pmc_envs_{which}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def mean(v):
    return sum(v) / len(v) if v else None


def is_batched_rollout(k):
    if not k.startswith("void rollout_pipe3_kernel<"):
        return False
    t = [x.strip() for x in k[k.index("<") + 1:k.index(">(")].split(",")]
    return len(t) >= 9 and t[4] == "true" and t[8] == "true"


tags = {"rollout_batched": is_batched_rollout, "square_tail": lambda k: "ns_square_tail_pair_kernel" in k,
        "iter_tail": lambda k: "ns_iter_tail_pair_kernel" in k, "T_quad": lambda k: "ns_T_quad_kernel" in k,
        "YZ_quad": lambda k: "ns_YZ_quad_kernel" in k}
out = {"kernel_src_sha": kernel_src_sha(), "kernels": {}}
try:
    out["commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"]).decode().strip()
except Exception:
    out["commit"] = None
for which in ("fetch", "write", "l2", "sq"):
    agg = load(which)
    for tag, pred in tags.items():
        ks = [k for k in agg if pred(k)]
        if not ks:
            continue
        k = max(ks, key=lambda k: sum(len(v) for v in agg[k].values()))
        e = out["kernels"].setdefault(tag, {"kernel": k[:120]})
        for c, v in agg[k].items():
            e[c] = mean(v)
        e.setdefault("dispatches", {})[which] = max(len(v) for v in agg[k].values())
ro = out["kernels"].get("rollout_batched", {})
if "FETCH_SIZE" in ro and "WRITE_SIZE" in ro:
    out["traffic_bytes_per_launch"] = int(round((2.0 * ro["FETCH_SIZE"] + ro["WRITE_SIZE"]) * 1024))
for tag in ("square_tail", "iter_tail", "T_quad", "YZ_quad"):
    e = out["kernels"].get(tag)
    if not e:
        continue
    d = {}
    if e.get("TCC_HIT_sum") is not None and e.get("TCC_MISS_sum") is not None and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    if e.get("SQ_WAVE_CYCLES"):
        d["wait_inst_any_share"] = e.get("SQ_WAIT_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
    if e.get("SQ_BUSY_CU_CYCLES") and e.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        d["mfma_busy_share_of_simd_busy"] = (e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (e["SQ_BUSY_CU_CYCLES"] * 4 / 1024)
    e["derived"] = d
out["command"] = ("rocprofv3 --pmc <one counter set per pass> -- python3 bench.py --config envs --steps 30 --warmup 5 --no-closed-loop "
                  "(scripts/profile_bench.sh)")
json.dump(out, open(os.path.join(root, "profiles", f"{RND}_bench_envs_pmc_summary.json"), "w"), indent=1)
print(json.dumps({"traffic_bytes_per_launch": out.get("traffic_bytes_per_launch"),
                  **{k: v.get("derived") for k, v in out["kernels"].items() if v.get("derived")}}, indent=1))
