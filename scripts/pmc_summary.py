#!/usr/bin/env python
"""Per-kernel means of the separate rocprofv3 --pmc passes over bench.py -> profiles/<round>_bench_pmc_*.csv and
profiles/<round>_bench_pmc_summary.json (what bench.py attaches as roofline.traffic / roofline.counters /
roofline_gemm.counters).

  fetch / write : FETCH_SIZE, WRITE_SIZE in KB per dispatch; FETCH_SIZE is doubled for the rollout's 16-B/lane streaming
                  reads (MI355X_MICROARCH.md, HBM section); the fabric counters include Infinity-Cache hits
  sq_a / sq_b   : SQ counters (quad-cycles for *_CYCLES / WAIT_* / ACTIVE_INST_*; SQ_VALU_MFMA_BUSY_CYCLES in cycles) and
                  GRBM_GUI_ACTIVE (cycles) -- effective shader clock = GRBM_GUI_ACTIVE / dispatch duration

usage: pmc_summary.py [gpurun_out dir] [round tag, default r04]"""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out")
RND = sys.argv[2] if len(sys.argv) > 2 else "r06"
N_LOCAL, H = 65536, 32
CU, SIMD = 256, 1024
sys.path.insert(0, root)
from bench import KERNEL_SRCS, kernel_src_sha  # noqa: E402  (ONE list: bench.py decides staleness with the same hash)

# cycles a SATURATED VALU pipe needs per wave-instruction, measured with the same counters on a known load
# (scripts/probe/valu_calib.hip -> profiles/r04_valu_calib.json: 8 independent v_fma_f32 on 8 waves per SIMD)
try:
    _cal = json.load(open(os.path.join(root, "profiles", "r04_valu_calib.json")))
    SAT_CYCLES_PER_VALU_INST = _cal["saturated_pipe_cycles_per_inst"]
    # a transcendental in the rollout's mix (7 fma + 1 rsq, 8 waves per SIMD): 8 x (cycles per instruction of the mix) - 7 x saturated
    _mix = [r for r in _cal["rows"] if r["stream"] == "7 fma + 1 rsq" and r["waves_per_simd"] == 8][0]
    TRANS_CYCLES = 8.0 * _mix["pipe_cycles_per_inst_measured"] - 7.0 * SAT_CYCLES_PER_VALU_INST
except Exception:
    SAT_CYCLES_PER_VALU_INST = TRANS_CYCLES = None


def load(which):
    """-> {kernel: {counter: [values per dispatch]}}, {kernel: [duration ns per dispatch] or []}"""
    files = glob.glob(os.path.join(src, f"pmc_{which}", "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for fn in files:
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[k][r.get("Dispatch_Id", len(dur[k]))] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, {k: list(v.values()) for k, v in dur.items()}


def mean(v):
    return sum(v) / len(v) if v else None


def pick(agg, pred):
    ks = [k for k in agg if pred(k)]
    return max(ks, key=lambda k: sum(len(v) for v in agg[k].values())) if ks else None


def rollout_targs(k):
    """template arguments of `void rollout_pipe3_kernel<DISC1, ROLL, CH, GROUPS, BATCHED, ONLY, ONLY_WAVES, STATS, REC, ...>(...)`"""
    if not k.startswith("void rollout_pipe3_kernel<"):
        return None
    return [t.strip() for t in k[k.index("<") + 1:k.index(">(")].split(",")]


def is_rollout_rec(k):   # the in-step variant: REC (9th template argument) true
    t = rollout_targs(k)
    return t is not None and len(t) >= 9 and t[8] == "true"


def is_rollout_plain(k):
    t = rollout_targs(k)
    return t is not None and not (len(t) >= 9 and t[8] == "true")


def is_gemm(k):
    return "noise_gemm_kernel" in k


out = {"kernel_src_sha": kernel_src_sha(), "n_local": N_LOCAL}
try:
    out["commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"]).decode().strip()
except Exception:
    out["commit"] = None

per_kernel = {}
for which in ("fetch", "write", "sq_a", "sq_b"):
    agg, dur = load(which)
    if not agg:
        continue
    counters = sorted({c for k in agg for c in agg[k]})
    with open(os.path.join(root, "profiles", f"{RND}_bench_pmc_{which}.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatches", "Duration_ns_mean"] + [c + "_mean" for c in counters])
        for k in sorted(agg, key=lambda k: -sum(sum(v) for v in agg[k].values())):
            n = max(len(v) for v in agg[k].values())
            d = mean(dur.get(k, []))
            w.writerow([k, n, f"{d:.0f}" if d else ""] + [f"{mean(agg[k][c]):.3f}" if agg[k].get(c) else "" for c in counters])
    for tag, pred in (("rollout_in_step", is_rollout_rec), ("rollout_standalone", is_rollout_plain), ("noise_gemm", is_gemm),
                      ("finalize_stream", lambda k: "ns_finalize_stream_kernel" in k)):
        k = pick(agg, pred)
        if k is None:
            continue
        e = per_kernel.setdefault(tag, {"kernel": k})
        for c, v in agg[k].items():
            e[c] = mean(v)
        if dur.get(k):
            e.setdefault("duration_us_under_pmc", {})[which] = mean(dur[k]) / 1e3
        e.setdefault("dispatches", {})[which] = max(len(v) for v in agg[k].values())

for tag, e in per_kernel.items():
    d = {}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e and tag.startswith("rollout"):
        d["traffic_bytes_per_launch"] = int(round((2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024))
    # SQ_BUSY_CU_CYCLES: quad-cycles summed over the SIMDs -> cycles a SIMD was busy during the dispatch.  (GRBM_GUI_ACTIVE of
    # a per-dispatch counter pass spans the profiler's whole sampling window -- 3e5 cycles around a 3 us copy -- and is NOT
    # the kernel's duration; it is kept in the csv but not used.)
    busy = e["SQ_BUSY_CU_CYCLES"] * 4 / SIMD if "SQ_BUSY_CU_CYCLES" in e else None
    du = (e.get("duration_us_under_pmc") or {})
    du = du.get("sq_b") or du.get("sq_a")
    if busy:
        d["busy_cycles_per_simd"] = busy
        if du:
            d["duration_us_under_pmc"] = du
            d["effective_clock_GHz_under_pmc"] = busy / (du * 1e3)   # lower bound: the launch ramp is inside `du`
    if "SQ_WAVE_CYCLES" in e:
        wc = e["SQ_WAVE_CYCLES"]
        d["wave_cycle_shares"] = {"active_inst_any": e["SQ_ACTIVE_INST_ANY"] / wc, "active_inst_valu": e["SQ_ACTIVE_INST_VALU"] / wc,
                                  "wait_inst_any": e["SQ_WAIT_INST_ANY"] / wc, "wait_inst_lds": e["SQ_WAIT_INST_LDS"] / wc,
                                  "wait_any": e["SQ_WAIT_ANY"] / wc}
        d["cycles_per_wave"] = wc * 4 / e["SQ_WAVES"]
        d["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
        if tag.startswith("rollout"):
            d["valu_wave_insts_per_64_samples_per_step"] = e["SQ_INSTS_VALU"] / (N_LOCAL / 64) / H
        # VALU-pipe utilisation, CALIBRATED.  SQ_ACTIVE_INST_VALU is a per-WAVE quad-cycle count summed
        # over co-resident waves: it reads exactly 4 "cycles" per instruction whatever the pipe does (profiles/r04_valu_calib.json:
        # 4.00 at every load; r03's formula read 1.79 for a saturated pipe) and is not pipe time.  What is: instructions per SIMD x
        # the measured cycles per instruction of a saturated pipe (2.24 on gfx950 at ~2.0 GHz; a transcendental holds it ~4.5x
        # as long, so this is a LOWER bound of the pipe's busy share for a stream with v_rsq / v_sqrt / v_log / v_rcp in it).
        insts_simd = e["SQ_INSTS_VALU"] / SIMD
        d["valu_insts_per_simd"] = insts_simd
        if busy and SAT_CYCLES_PER_VALU_INST:
            valu = insts_simd * SAT_CYCLES_PER_VALU_INST
            d["saturated_pipe_cycles_per_valu_inst"] = SAT_CYCLES_PER_VALU_INST
            d["valu_pipe_cycles_per_simd"] = valu
            d["valu_pipe_util"] = valu / busy
            d["cycles_per_valu_inst_achieved"] = busy / insts_simd
            if du:
                d["valu_pipe_floor_us_at_that_clock"] = valu / (busy / du)
        if busy and SAT_CYCLES_PER_VALU_INST and "SQ_INSTS_VALU_TRANS_F32" in e:
            # with the transcendentals at their measured pipe time (13 cycles each in a 7 + 1 mix, against 2.24)
            tr = e["SQ_INSTS_VALU_TRANS_F32"] / SIMD
            pipe = (insts_simd - tr) * SAT_CYCLES_PER_VALU_INST + tr * TRANS_CYCLES
            d["trans_insts_per_simd"] = tr
            d["trans_pipe_cycles_per_inst"] = TRANS_CYCLES
            d["valu_and_trans_pipe_cycles_per_simd"] = pipe
            d["valu_and_trans_pipe_util"] = pipe / busy
            if du:
                d["valu_and_trans_pipe_floor_us_at_that_clock"] = pipe / (busy / du)
            if tag.startswith("rollout"):
                d["trans_wave_insts_per_64_samples_per_step"] = e["SQ_INSTS_VALU_TRANS_F32"] / (N_LOCAL / 64) / H
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and busy and e["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
        mf = e["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMD   # cycles a SIMD's matrix pipe was busy
        d["mfma_pipe_cycles_per_simd"] = mf
        d["mfma_util"] = mf / busy
        d["mfma_insts_per_wave"] = e["SQ_INSTS_MFMA"] / e["SQ_WAVES"] if e.get("SQ_WAVES") else None
        d["mfma_busy_cycles_per_inst"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_INSTS_MFMA"]
    if "SQ_INSTS_LDS" in e and tag.startswith("rollout"):
        d["lds_wave_insts_per_64_samples_per_step"] = e["SQ_INSTS_LDS"] / (N_LOCAL / 64) / H
    e["derived"] = d

out["kernels"] = per_kernel
if "rollout_in_step" in per_kernel and "traffic_bytes_per_launch" in per_kernel["rollout_in_step"]["derived"]:
    out["traffic_bytes_per_launch"] = per_kernel["rollout_in_step"]["derived"]["traffic_bytes_per_launch"]
    out["kernel"] = "rollout_pipe3_kernel<..., REC = true> (the in-step variant)"
out["note"] = ("N_local=65536; traffic = 2 x FETCH_SIZE (gfx950 16-B/lane correction) + WRITE_SIZE, fabric counters include "
               "Infinity-Cache hits; SQ *_CYCLES / WAIT_* / ACTIVE_INST_* are quad-cycles summed over all waves")
out["command"] = ("rocprofv3 --pmc <one counter set per pass> -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline "
                  "--no-closed-loop --no-info-leg   (scripts/profile_bench.sh)")
json.dump(out, open(os.path.join(root, "profiles", f"{RND}_bench_pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: v.get("derived") for k, v in per_kernel.items()}, indent=1))
