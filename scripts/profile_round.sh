#!/bin/bash
# Everything profiles/<round>_* holds, in one GPU call: scripts/profile_bench.sh (bench line, rocprofv3 stats, PMC passes, phase
# times, kernel table) + every config + the probes.  Run on the GPU box from the repo root; outputs land in gpurun_out/.
cd "$(dirname "$0")/.."
bash scripts/profile_bench.sh > gpurun_out/profile_bench.log 2>&1
( bash scripts/all_configs.sh; python scripts/env_batched_step.py --envs 1 32 64 ) > gpurun_out/all_configs.log 2>&1
for p in valu_probe rbody_probe xcd_chain_probe; do timeout 120 ./scripts/probe/$p > gpurun_out/$p.log 2>&1; done
LAB_STAGES=1 timeout 300 ./scripts/probe/rollout_lab 8192 32768 65536 131072 1048576 > gpurun_out/rollout_lab.log 2>&1
timeout 60 ./scripts/probe/pipe_timeline > gpurun_out/pipe_timeline.log 2>&1
tail -3 gpurun_out/all_configs.log; grep -c . gpurun_out/rollout_lab.log gpurun_out/valu_probe.log
