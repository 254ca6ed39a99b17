#!/bin/bash
# Everything profiles/ holds for one round, in one GPU call: the unprofiled bench line, the rocprofv3 kernel stats of the
# same command, the separate --pmc passes (FETCH_SIZE, WRITE_SIZE, two SQ passes; counters are never combined with
# traces, every pass has the program itself right after `--`), the phase times and the stand-alone kernel table.
# Run on the GPU box from the repo root; outputs land in gpurun_out/.   usage: scripts/profile_bench.sh [round tag, default r05]
R="$(cd "$(dirname "$0")/.." && pwd)"
RND="${1:-r06}"
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" > "$R/gpurun_out/bench_line.json" 2> "$R/gpurun_out/bench_line.err"
rm -rf "$R/gpurun_out/prof_stats" "$R/gpurun_out/pmc_fetch" "$R/gpurun_out/pmc_write" "$R/gpurun_out/pmc_sq_a" "$R/gpurun_out/pmc_sq_b"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o bench -- python3 "$R/bench.py" \
    > "$R/gpurun_out/bench_line_under_rocprof.json" 2> "$R/gpurun_out/prof_stats.err"
PMC_ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-closed-loop --no-info-leg --no-sweep"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch" -o bench -- python3 "$R/bench.py" $PMC_ARGS \
    > /dev/null 2> "$R/gpurun_out/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write" -o bench -- python3 "$R/bench.py" $PMC_ARGS \
    > /dev/null 2> "$R/gpurun_out/pmc_write.err"
# SQ pass A (8 SQ slots + GRBM): where the waves' cycles go -- issue (ACTIVE_INST_*), issue stalls (WAIT_INST_*), parked (WAIT_ANY)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE \
    --output-format csv -d "$R/gpurun_out/pmc_sq_a" -o bench -- python3 "$R/bench.py" $PMC_ARGS > /dev/null 2> "$R/gpurun_out/pmc_sq_a.err"
# SQ pass B: matrix-pipe busy cycles (MFMA utilisation of the noise GEMM), LDS instruction counts / conflicts
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE \
    --output-format csv -d "$R/gpurun_out/pmc_sq_b" -o bench -- python3 "$R/bench.py" $PMC_ARGS > /dev/null 2> "$R/gpurun_out/pmc_sq_b.err"
# BASELINE configs[4] (bench.py --config envs): its line and the kernel stats of the same command
python3 "$R/bench.py" --config envs > "$R/gpurun_out/bench_envs_line.json" 2> /dev/null
rm -rf "$R/gpurun_out/prof_envs"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_envs" -o envs -- python3 "$R/bench.py" --config envs --steps 50 --warmup 5 \
    > /dev/null 2> "$R/gpurun_out/prof_envs.err"
# config [4]'s counter passes: the batched rollout's HBM traffic and the batched Sigma chain's launches
ENVS_ARGS="--config envs --steps 30 --warmup 5 --no-closed-loop"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVES"; do
    set -- $pass
    name=$1; shift
    rm -rf "$R/gpurun_out/pmc_envs_$name"
    rocprofv3 --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc_envs_$name" -o envs -- python3 "$R/bench.py" $ENVS_ARGS \
        > /dev/null 2> "$R/gpurun_out/pmc_envs_$name.err"
done
cd "$R"
python3 scripts/pmc_summary_envs.py gpurun_out "$RND" > gpurun_out/pmc_summary_envs.log 2>&1
python3 scripts/phase_times.py > gpurun_out/phase_times.log 2>/dev/null
python3 scripts/kbench.py > gpurun_out/kbench.log 2>/dev/null
python3 scripts/pmc_summary.py gpurun_out "$RND" > gpurun_out/pmc_summary.log 2>&1
# gpurun merges at most 64 MiB back: the per-dispatch counter dumps (4 x ~14 MB) and the kernel trace stay on the box, their
# per-kernel means (profiles/<round>_bench_pmc_*.csv, *_summary.json: written above) and the kernel stats travel
mkdir -p gpurun_out/profiles_out
cp profiles/"$RND"_bench_pmc_* profiles/"$RND"_bench_envs_pmc_summary.json gpurun_out/profiles_out/
cp gpurun_out/prof_stats/bench_kernel_stats.csv gpurun_out/profiles_out/"$RND"_bench_kernel_stats.csv
find gpurun_out/prof_envs -name "*kernel_stats.csv" -exec cp {} gpurun_out/profiles_out/"$RND"_bench_envs_kernel_stats.csv \;
cp gpurun_out/bench_envs_line.json gpurun_out/profiles_out/"$RND"_bench_envs_line.json
cp gpurun_out/bench_line.json gpurun_out/profiles_out/"$RND"_bench_line.json
cp gpurun_out/bench_line_under_rocprof.json gpurun_out/profiles_out/"$RND"_bench_line_under_rocprof.json
cp gpurun_out/phase_times.log gpurun_out/profiles_out/"$RND"_phase_times.log
cp gpurun_out/kbench.log gpurun_out/profiles_out/"$RND"_kbench.log
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b gpurun_out/prof_stats gpurun_out/prof_envs gpurun_out/pmc_envs_*
ls gpurun_out/profiles_out
tail -c 1500 gpurun_out/bench_line.json
tail -5 gpurun_out/pmc_summary.log
