#!/bin/bash
# GPU session 4 of round 6: DCN_HL_MIN_K (the 1 x 1 convolutions fed by 128 - 1023 channels on the hl32 kernels) -- parity of all five
# configs under the new default, then same-box A/B of the step: default (128) vs the round-5 thresholds (DCN_HL_MIN_K=1024).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_round4.py tests/test_gpu_round6.py -m gpu -q -x > gpurun_out/r6g_parity_min_k.log 2>&1; tail -4 gpurun_out/r6g_parity_min_k.log
{
for rep in 1 2; do
for wl in config2 config1 config5 config4; do
for sep in "" "--separate-forwards"; do
  if [ "$wl" != "config1" ] && [ "$wl" != "config2" ] && [ -n "$sep" ]; then continue; fi
  for k in 128 1024; do
    DCN_HL_MIN_K=$k timeout 600 python bench.py --workload $wl $sep --no-variants --cpu-baseline-steps 0 --profile-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep  %-8s %-20s DCN_HL_MIN_K=%-5s %8.2f images/s  %8.3f ms/step' % ('$wl', '$sep' or 'forward_pair', '$k', d['value'], d['ms_per_step']))"
  done
done; done; done
} 2>&1 | tee gpurun_out/r6g_ab_hl_min_k.txt
{
for n in 8 2; do
echo "# r34 n=$n default"; python tools/conv_bench.py --mode hl --n $n --reps 15 --relu-x 2>&1 | grep -v amdgpu.ids | tail -14
echo "# r34 n=$n DCN_GEMM_HL=2 DCN_WGRAD_HL=2 DCN_WGRAD_HLR=2"; DCN_GEMM_HL=2 DCN_WGRAD_HL=2 DCN_WGRAD_HLR=2 python tools/conv_bench.py --mode hl --n $n --reps 15 --relu-x 2>&1 | grep -v amdgpu.ids | tail -14
done
} > gpurun_out/r6g_conv_per_layer_default_vs_forced.txt 2>&1
cut -c1-190 gpurun_out/r6g_conv_per_layer_default_vs_forced.txt
