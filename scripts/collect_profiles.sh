#!/bin/bash
# Copy what scripts/profile_round.sh left in gpurun_out/ into profiles/<round>_* (run locally after the gpurun call).
R=${1:-r02}
cd "$(dirname "$0")/.."
python3 scripts/pmc_summary.py gpurun_out $R > /dev/null
cp gpurun_out/bench_line.json profiles/${R}_bench_line.json
cp gpurun_out/bench_line_under_rocprof.json profiles/${R}_bench_line_under_rocprof.json
cp gpurun_out/prof_stats/bench_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp gpurun_out/phase_times.log profiles/${R}_phase_times.log
cp gpurun_out/kbench.log profiles/${R}_kbench.log
grep -v "amdgpu.ids" gpurun_out/all_configs.log > profiles/${R}_all_configs.log
for p in valu_probe rbody_probe xcd_chain_probe rollout_lab pipe_timeline; do cp gpurun_out/$p.log profiles/${R}_$p.log; done
ls -la profiles/${R}_*
