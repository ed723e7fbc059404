#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "# default"; python tools/conv_bench.py --mode hl --n 8 --only "layer3 3x3" --kinds fwd,dgrad --reps 30 --relu-x 2>&1 | grep layer3
echo "# DCN_GEMM_HL_ROWS=320 (120 tiles, one per CU)"; DCN_GEMM_HL_ROWS=320 python tools/conv_bench.py --mode hl --n 8 --only "layer3 3x3" --kinds fwd,dgrad --reps 30 --relu-x 2>&1 | grep layer3
echo "# DCN_GEMM_HL_ROWS=320 DCN_GEMM_SK=240 (K split in two halves)"; DCN_GEMM_HL_ROWS=320 DCN_GEMM_SK=240 python tools/conv_bench.py --mode hl --n 8 --only "layer3 3x3" --kinds fwd,dgrad --reps 30 --relu-x 2>&1 | grep layer3
echo "# DCN_GEMM_HL_ROWS=320 DCN_GEMM_SK=256"; DCN_GEMM_HL_ROWS=320 DCN_GEMM_SK=256 python tools/conv_bench.py --mode hl --n 8 --only "layer3 3x3" --kinds fwd,dgrad --reps 30 --relu-x 2>&1 | grep layer3
echo "# DCN_GEMM_HL_ROWS=256 DCN_GEMM_SK=256 (150 tiles stream-K'd over 256 workgroups)"; DCN_GEMM_HL_ROWS=256 DCN_GEMM_SK=256 python tools/conv_bench.py --mode hl --n 8 --only "layer3 3x3" --kinds fwd,dgrad --reps 30 --relu-x 2>&1 | grep layer3
echo "# DCN_GEMM_HL_ROWS=192 DCN_GEMM_SK=256"; DCN_GEMM_HL_ROWS=192 DCN_GEMM_SK=256 python tools/conv_bench.py --mode hl --n 8 --only "layer3 3x3" --kinds fwd,dgrad --reps 30 --relu-x 2>&1 | grep layer3
} 2>&1 | tee gpurun_out/r6h_layer3_ksplit.txt | cut -c1-200
