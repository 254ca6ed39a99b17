#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc pass over scripts/probe/valu_calib: per (instruction stream, waves per
SIMD) what the SQ counters read for a VALU pipe whose load is known -> profiles/<round>_valu_calib.json.

  busy cycles per SIMD      = SQ_BUSY_CU_CYCLES x 4 / 1024      (quad-cycles summed over the SIMDs)
  pipe cycles per inst      = busy cycles per SIMD / (SQ_INSTS_VALU / 1024)   -- the MEASURED issue interval of the pipe
  active quad-cycles / inst = SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU         -- what pmc_summary.py r03 called "pipe cycles"
  old "valu_pipe_util"      = SQ_ACTIVE_INST_VALU x 4 / 1024 / busy           -- r03's formula, for comparison

usage: valu_calib_summary.py [dir with *counter_collection.csv] [round tag]"""
import collections
import csv
import glob
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "valu_calib")
RND = sys.argv[2] if len(sys.argv) > 2 else "r04"
SIMD = 1024
KIND = {0: "dependent v_fma_f32", 1: "8 independent v_fma_f32", 2: "7 fma + 1 rsq"}

agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for fn in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"calib<(\d+), (\d+)>", r["Kernel_Name"])
        if not m:
            continue
        k = (int(m.group(1)), int(m.group(2)))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[k][r.get("Dispatch_Id")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])

rows = []
for (kind, w) in sorted(agg):
    e = {c: sum(v) / len(v) for c, v in agg[(kind, w)].items()}
    d = sorted(dur[(kind, w)].values())
    du = d[len(d) // 2] / 1e3 if d else None
    busy = e["SQ_BUSY_CU_CYCLES"] * 4 / SIMD
    insts = e["SQ_INSTS_VALU"] / SIMD
    row = {"stream": KIND[kind], "waves_per_simd": w, "valu_insts_per_simd": insts, "busy_cycles_per_simd": busy,
           "pipe_cycles_per_inst_measured": busy / insts,
           "active_inst_valu_quadcycles_x4_per_inst": e["SQ_ACTIVE_INST_VALU"] * 4 / e["SQ_INSTS_VALU"],
           "r03_formula_valu_pipe_util": e["SQ_ACTIVE_INST_VALU"] * 4 / SIMD / busy,
           "duration_us_under_pmc": du, "effective_clock_GHz": busy / (du * 1e3) if du else None}
    if "SQ_WAVE_CYCLES" in e:
        wc = e["SQ_WAVE_CYCLES"]
        row["wave_cycle_shares"] = {"active_inst_valu": e["SQ_ACTIVE_INST_VALU"] / wc,
                                    "wait_inst_any": e.get("SQ_WAIT_INST_ANY", 0.0) / wc, "wait_any": e.get("SQ_WAIT_ANY", 0.0) / wc}
    rows.append(row)

# the saturated pipe: the smallest measured cycles per instruction over the independent-FMA runs
sat = min((r["pipe_cycles_per_inst_measured"] for r in rows if r["stream"] == KIND[1]), default=None)
for r in rows:
    r["pipe_util_calibrated"] = sat / r["pipe_cycles_per_inst_measured"] if sat else None
out = {"saturated_pipe_cycles_per_inst": sat, "rows": rows,
       "note": "pipe_util_calibrated = (cycles per VALU instruction of the saturated pipe: 8 independent v_fma_f32, best W) / "
               "(busy cycles per SIMD / VALU instructions per SIMD).  r03_formula_valu_pipe_util is the r03 figure "
               "(SQ_ACTIVE_INST_VALU x 4 per SIMD / busy): SQ_ACTIVE_INST_VALU counts per-WAVE quad-cycles, summed over "
               "co-resident waves, so it is not pipe time."}
json.dump(out, open(os.path.join(root, "profiles", f"{RND}_valu_calib.json"), "w"), indent=1)
for r in rows:
    print(f'{r["stream"]:26s} W={r["waves_per_simd"]}  cyc/inst {r["pipe_cycles_per_inst_measured"]:6.2f}  active*4/inst '
          f'{r["active_inst_valu_quadcycles_x4_per_inst"]:5.2f}  r03-util {r["r03_formula_valu_pipe_util"]:5.2f}  '
          f'calibrated util {r["pipe_util_calibrated"]:5.2f}  clk {r["effective_clock_GHz"] or 0:4.2f} GHz')
print("saturated pipe:", sat, "cycles per VALU wave-instruction")
