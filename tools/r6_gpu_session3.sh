#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for n in 4 2; do
echo "# r50 n=$n default eligibility"; python tools/conv_bench.py --shapes r50 --mode hl --n $n --reps 10 --relu-x 2>&1 | tail -14
echo "# r50 n=$n DCN_GEMM_HL=2 DCN_WGRAD_HL=2"; DCN_GEMM_HL=2 DCN_WGRAD_HL=2 python tools/conv_bench.py --shapes r50 --mode hl --n $n --reps 10 --relu-x 2>&1 | tail -14
done
} > gpurun_out/r6f_r50_hl_everywhere.txt 2>&1
cut -c1-200 gpurun_out/r6f_r50_hl_everywhere.txt
