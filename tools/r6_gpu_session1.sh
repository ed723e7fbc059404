#!/bin/bash
# GPU session 1 of round 6: the whole GPU suite (slow sweeps included) on the round-6 tree, smoke, the default bench line,
# the one-rank RCCL run of the bucketed schedule, and the kernel-time breakdown of a config-2 step.
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
DCN_RUN_SLOW=1 timeout 1500 python -m pytest tests -m gpu -q -x -rs --durations=15 > gpurun_out/r6b_slow_gpu_suite.log 2>&1
tail -5 gpurun_out/r6b_slow_gpu_suite.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r6b_smoke.log 2>&1; tail -2 gpurun_out/r6b_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6b_bench_default.json 2> gpurun_out/r6b_bench_default.err
tail -c 1500 gpurun_out/r6b_bench_default.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --no-variants --cpu-baseline-steps 0 > gpurun_out/r6b_bench_force_dist_1rank.json 2> gpurun_out/r6b_bench_force_dist_1rank.err
tail -c 700 gpurun_out/r6b_bench_force_dist_1rank.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r6b_prof" -o r6b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-variants --cpu-baseline-steps 0 --profile-steps 0 > "$GRAFT_REPO_ROOT/gpurun_out/r6b_prof_bench.json" 2>&1
cd "$GRAFT_REPO_ROOT"; find gpurun_out/r6b_prof -name "*kernel_stats*" | head; f=$(find gpurun_out/r6b_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" > gpurun_out/r6b_kernel_stats.csv
find gpurun_out/r6b_prof -type f ! -name "*stats*" -delete
