#!/bin/bash
# kernel-time breakdown (serial schedule) of the reference's own operating point: config 1 as a pair and as two calls
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in pair separate; do
  extra=""; [ $v = separate ] && extra="--separate-forwards"
  (cd /tmp && timeout 600 env DCN_BACKWARD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6_prof_c1_$v -- python $GRAFT_REPO_ROOT/bench.py --workload config1 $extra --steps 20 --warmup 5 --cpu-baseline-steps 0 --no-variants --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r6_prof_c1_$v.log 2>&1)
  python tools/stats_summary.py gpurun_out/r6_prof_c1_$v > gpurun_out/r6_kernel_stats_config1_$v.txt 2>&1
  head -14 gpurun_out/r6_kernel_stats_config1_$v.txt | cut -c1-150
done
find gpurun_out -name "*.db" -size +20M -delete 2>/dev/null
