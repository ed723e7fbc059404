#!/bin/bash
# Round-5 GPU-box sessions (run through gpurun from the repo root): bash tools/gpu_r5_session.sh <tag> [steps...]
# Steps of its own first; anything else is handed to tools/gpu_r4_session.sh (bench, prof, pmc, tests, smoke, ab, convn2, convn8 ...).
tag=${1:-r5}; shift
steps=${*:-tests5 sweep}
mkdir -p gpurun_out
export TMPDIR=/tmp
for s in $steps; do
  echo "=== $s $(date +%T)"
  case $s in
    tests5)    timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q -x -p no:cacheprovider --durations=5 > gpurun_out/${tag}_pytest_r5.log 2>&1; tail -12 gpurun_out/${tag}_pytest_r5.log | cut -c1-300 ;;
    sweep)     timeout 600 python tools/hlx_sweep.py --n ${SWEEP_N:-1,2,4,8} ${SWEEP_ARGS:-} 2>&1 | grep -v "Warn\|amdgpu.ids" | cut -c1-250 | tee gpurun_out/${tag}_hlx_sweep.txt ;;
    sweepnarrow) timeout 300 python tools/hlx_sweep.py --n 2,4,8 --narrow --only layer2 --variants default,old,2:1,2:2,2:3 2>&1 | grep -v "Warn\|amdgpu.ids" | cut -c1-250 | tee gpurun_out/${tag}_hlx_sweep_narrow.txt ;;
    crashn2)   # which launch of the forced-hl N = 2 table faults (profiles/r4a_conv_per_layer_n2.txt ended in a memory access fault)
               for kind in fwd dgrad wgrad; do echo "--- layer4.0 $kind, DCN_GEMM_HL=2 DCN_WGRAD_HL=2 DCN_GEMM_HLX=0, N = 2" | tee -a gpurun_out/${tag}_crashn2.txt
                 timeout 120 env DCN_GEMM_HL=2 DCN_WGRAD_HL=2 DCN_GEMM_HLX=0 python tools/conv_bench.py --mode hl --n 2 --only "layer4.0" --kinds $kind --reps 5 2>&1 | grep -v "Warn\|amdgpu.ids" | cut -c1-200 | tail -4 | tee -a gpurun_out/${tag}_crashn2.txt; done ;;
    wgradn2)   # weight gradients at two images: fp32-operand kernel vs the hl32 kernel without its M >= 16384 gate
               for hl in 1 2; do echo "--- DCN_WGRAD_HL=$hl, N = ${WG_N:-2}" | tee -a gpurun_out/${tag}_wgrad_n2.txt
                 timeout 200 env DCN_WGRAD_HL=$hl python tools/conv_bench.py --mode hl --n ${WG_N:-2} --only "layer" --kinds wgrad --x-direct --reps 20 2>&1 | grep -v "Warn\|amdgpu.ids" | grep "layer3\|layer4" | cut -c1-200 | tee -a gpurun_out/${tag}_wgrad_n2.txt; done ;;
    convn)     # per-layer table through the C ABI at N = ${CONV_N:-2} images, the kernels the library picks by default
               timeout 400 python tools/conv_bench.py --mode hl --n ${CONV_N:-2} --x-direct --reps 30 --relu-x 2>&1 | grep -v "Warn\|amdgpu.ids" | cut -c1-170 | tee gpurun_out/${tag}_conv_per_layer_n${CONV_N:-2}.txt ;;
    sq)        # SQ counters of the hl32 kernels at N = 8 (two --pmc passes, each with --kernel-trace only)
               i=0; for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do i=$((i+1))
                 (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_sq_$i -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --mode hl --n 8 --kinds ${SQ_KINDS:-fwd,dgrad,wgrad} --x-direct --no-split --only "${SQ_ONLY:-3x3 d}" --reps 5 --relu-x > $GRAFT_REPO_ROOT/gpurun_out/${tag}_sq_$i.log 2>&1); done
               python tools/sq_summary.py gpurun_out/${tag}_hl_sq_counters.txt "rocprofv3 --kernel-trace --pmc <two passes> -- python tools/conv_bench.py --mode hl --n 8 --kinds fwd,dgrad,wgrad --x-direct --no-split --only '3x3 d' --reps 5 --relu-x" gpurun_out/${tag}_sq_1 gpurun_out/${tag}_sq_2; cat gpurun_out/${tag}_hl_sq_counters.txt | cut -c1-120 ;;
    fp32all)   # the exact-fp32 MFMA arithmetic beside every split-fp16 row of the results table
               for w in config1 config2 config3 config4 config5; do for sep in "" "--separate-forwards"; do
                 [ -n "$sep" ] && [ "$w" != "config1" ] && [ "$w" != "config2" ] && continue
                 timeout 300 python bench.py --workload $w $sep --conv-mode fp32 --no-variants --cpu-baseline-steps 0 --profile-steps 0 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 mode %s %s: %.1f images/s  %.2f ms/step' % ('$w', '$sep' or 'pair', d['value'], d['ms_per_step']))" | tee -a gpurun_out/${tag}_fp32_mode.txt; done; done ;;
    dist1)     # the RCCL path with ONE rank (nobody to talk to: what the collective machinery costs a step)
               timeout 300 python bench.py --force-dist --no-variants --cpu-baseline-steps 0 --profile-steps 0 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_force_dist_1rank.json; python -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_force_dist_1rank.json').read()); print('force-dist, 1 rank: %.1f images/s  %.2f ms/step  %s' % (d['value'], d['ms_per_step'], {k: d.get(k) for k in ('allreduce_ms','communication') if k in d}))" ;;
    hlr)       # row-window weight-gradient kernel of the narrow 3 x 3 layers against the fp32-operand kernel, N = 8 and N = 2
               for nn in ${HLR_N:-8 2}; do for v in ${HLR_V:-1 0}; do echo "--- DCN_WGRAD_HLR=$v DCN_WGRAD_HLR_MIN_M=1 DCN_WGRAD_HLR_PAIRS=${DCN_WGRAD_HLR_PAIRS:-default}, N = $nn" | tee -a gpurun_out/${tag}_wgrad_hlr.txt
                 timeout 200 env DCN_WGRAD_HLR=$v DCN_WGRAD_HLR_MIN_M=1 python tools/conv_bench.py --mode hl --n $nn --only "3x3 " --kinds wgrad --x-direct --reps 20 2>&1 | grep -v "Warn\|amdgpu.ids" | grep "layer1\|layer2 3x3" | cut -c1-200 | tee -a gpurun_out/${tag}_wgrad_hlr.txt; done; done ;;
    *)         bash tools/gpu_r4_session.sh $tag $s ;;
  esac
done
find gpurun_out -name "*.db" -size +20M -delete 2>/dev/null
echo "=== r5 done $(date +%T)"
