export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout 900 env DCN_BACKWARD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3k_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline-steps 0 --no-variants --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r3k_prof_bench.log 2>&1); python tools/stats_summary.py gpurun_out/r3k_prof > gpurun_out/r3k_kernel_stats.txt 2>&1; head -45 gpurun_out/r3k_kernel_stats.txt | cut -c1-150
find gpurun_out -name "*.db" -size +20M -delete 2>/dev/null
