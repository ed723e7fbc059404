#!/bin/bash
# One GPU-box session (run through gpurun from the repo root): parity report, GPU test-suite, smoke, bench lines, profile.
# Usage: bash tools/gpu_session.sh <tag> [steps...]   steps: report tests smoke bench forcedist config3 configs prof pmc pmc3 prof3 convbench
#        A/B steps (same box, alternating): ab (stream-K completion x BN reduction), abenv (AB_ENV="NAME a b"), abw / wgab (wgrad tile),
#        wdab (wgrad deep prefetch), skab (stream-K threshold), sqw (SQ counters of the wgrad tiles), tests2b, parity
# Everything lands in gpurun_out/<tag>_*; nothing here reads /root/reference.
tag=${1:-r2}; shift
steps=${*:-report tests smoke bench forcedist config3 prof}
mkdir -p gpurun_out
export TMPDIR=/tmp
for s in $steps; do
  echo "=== $s $(date +%T)"
  case $s in
    report)    timeout 900 python tools/parity_report.py --json gpurun_out/${tag}_parity_report.json > gpurun_out/${tag}_parity_report.log 2>&1; tail -12 gpurun_out/${tag}_parity_report.log | cut -c1-600 ;;
    tests)     timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${tag}_pytest.log 2>&1; tail -40 gpurun_out/${tag}_pytest.log | cut -c1-400 ;;
    smoke)     timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1; tail -3 gpurun_out/${tag}_smoke.log ;;
    bench)     timeout 600 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err; tail -c 1500 gpurun_out/${tag}_bench_default.json; tail -5 gpurun_out/${tag}_bench_default.err ;;
    forcedist) timeout 600 env NCCL_DEBUG=VERSION python bench.py --force-dist --no-variants --cpu-baseline-steps 0 > gpurun_out/${tag}_bench_forcedist.log 2>&1; tail -c 1200 gpurun_out/${tag}_bench_forcedist.log ;;
    config3)   timeout 600 python bench.py --workload config3 --no-variants --cpu-baseline-steps 0 --steps 10 --warmup 3 > gpurun_out/${tag}_bench_config3.json 2> gpurun_out/${tag}_bench_config3.err; tail -c 1200 gpurun_out/${tag}_bench_config3.json ;;
    configs)   for c in config1 config4 config5; do timeout 600 python bench.py --workload $c --no-variants --cpu-baseline-steps 0 --steps 10 --warmup 3 > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err; head -c 300 gpurun_out/${tag}_bench_$c.json; echo; done ;;
    prof)      (cd /tmp && timeout 900 env DCN_BACKWARD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline-steps 0 --no-variants --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_bench.log 2>&1); python tools/stats_summary.py gpurun_out/${tag}_prof > gpurun_out/${tag}_kernel_stats.txt 2>&1; head -45 gpurun_out/${tag}_kernel_stats.txt ;;
    pmc)       for ctr in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$ctr -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-baseline-steps 0 --no-variants --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$ctr.log 2>&1); done; python tools/pmc_summary.py gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE gpurun_out/${tag}_hbm_counters.txt gpurun_out/${tag}_hbm_counters.json 2>&1 | tail -3; head -40 gpurun_out/${tag}_hbm_counters.txt ;;
    pmc3)      for ctr in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc3_$ctr -- python $GRAFT_REPO_ROOT/bench.py --workload config3 --steps 2 --warmup 1 --cpu-baseline-steps 0 --no-variants --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc3_$ctr.log 2>&1); done; DCN_PMC_WORKLOAD=config3 python tools/pmc_summary.py gpurun_out/${tag}_pmc3_FETCH_SIZE gpurun_out/${tag}_pmc3_WRITE_SIZE gpurun_out/${tag}_hbm_counters_config3.txt gpurun_out/${tag}_hbm_counters_config3.json 2>&1 | tail -3; grep -i "loss\|upsample\|fill" gpurun_out/${tag}_hbm_counters_config3.txt ;;
    newtests)  timeout 900 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_round2.py -m gpu -q -p no:cacheprovider > gpurun_out/${tag}_pytest_new.log 2>&1; tail -15 gpurun_out/${tag}_pytest_new.log | cut -c1-300 ;;
    convbench) timeout 600 python tools/conv_bench.py --mode f16 --n 16 --x-direct --json gpurun_out/${tag}_conv_per_layer.json > gpurun_out/${tag}_conv_per_layer.txt 2>&1; cat gpurun_out/${tag}_conv_per_layer.txt | cut -c1-170
               timeout 600 env DCN_GEMM_SK=0 python tools/conv_bench.py --mode f16 --n 16 --kinds fwd,dgrad --json gpurun_out/${tag}_conv_per_layer_nosk.json > gpurun_out/${tag}_conv_per_layer_nosk.txt 2>&1; cat gpurun_out/${tag}_conv_per_layer_nosk.txt | cut -c1-120 ;;
    prof3)     (cd /tmp && timeout 900 env DCN_BACKWARD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof3 -- python $GRAFT_REPO_ROOT/bench.py --workload config3 --steps 3 --warmup 1 --cpu-baseline-steps 0 --no-variants --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof3_bench.log 2>&1); python tools/stats_summary.py gpurun_out/${tag}_prof3 > gpurun_out/${tag}_kernel_stats_config3.txt 2>&1; grep -i "loss\|fill\|upsample\|total" gpurun_out/${tag}_kernel_stats_config3.txt ;;
    tests2b)   timeout 1200 python -m pytest tests/test_gpu_round2b.py -m gpu -q -s -p no:cacheprovider > gpurun_out/${tag}_pytest_2b.log 2>&1; tail -30 gpurun_out/${tag}_pytest_2b.log | cut -c1-300 ;;
    skab)      # stream-K threshold A/B (stage times stream-K must save to be chosen), per layer at N = 8 and on the step
               for g in 20 10 5; do echo "--- DCN_GEMM_SK_MIN_GAIN=$g, N = 8" | tee -a gpurun_out/${tag}_skab.txt
                 timeout 300 env DCN_GEMM_SK_MIN_GAIN=$g python tools/conv_bench.py --mode f16 --n 8 --kinds fwd,dgrad --only layer --reps 20 2>&1 | grep -v "Warn\|amdgpu.ids" | cut -c1-150 | tee -a gpurun_out/${tag}_skab.txt
               done
               for rep in 1 2; do for g in 20 10 5; do
                 timeout 300 env DCN_GEMM_SK_MIN_GAIN=$g python bench.py --steps 20 --warmup 5 --cpu-baseline-steps 0 --no-variants --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('absk sk_min_gain=$g rep=$rep  %.1f images/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/${tag}_skab.txt
               done; done ;;
    abenv)     # same-box A/B of one environment switch on the whole step: AB_ENV="NAME a b" (default: residual add deferred 1 | in the epilogue 0)
               set -- ${AB_ENV:-DCN_DEFER_RESIDUAL_ADD 1 0}; name=$1; shift
               for rep in 1 2 3; do for v in "$@"; do
                 timeout 300 env $name=$v python bench.py --steps 20 --warmup 5 --cpu-baseline-steps 0 --no-variants --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abenv $name=$v rep=$rep  %.1f images/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/${tag}_abenv.txt
               done; done ;;
    sqw)       # SQ counters of the wgrad kernel on the layer-4 3x3 convolution, N = 8: 256-channel / 8-wavefront tile vs 128-channel
               for t in 0 128; do i=0; for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do i=$((i+1))
                 (cd /tmp && timeout 300 env DCN_WGRAD_TILE=$t rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_sqw_${t}_$i -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --mode f16 --n 8 --kinds wgrad --x-direct --no-split --only "layer4 3x3" --reps 5 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_sqw_${t}_$i.log 2>&1)
               done
               python tools/sq_summary.py gpurun_out/${tag}_wgrad_sq_counters_tile$t.txt "rocprofv3 --kernel-trace --pmc <two passes> -- DCN_WGRAD_TILE=$t python tools/conv_bench.py --mode f16 --n 8 --kinds wgrad --x-direct --no-split --only 'layer4 3x3' --reps 5  (0 = default: 256-channel tile on 8 wavefronts; 128 = the 128-channel / 4-wavefront tile)" gpurun_out/${tag}_sqw_${t}_1 gpurun_out/${tag}_sqw_${t}_2; cat gpurun_out/${tag}_wgrad_sq_counters_tile$t.txt
               done ;;
    wdab)      # wgrad deep-prefetch A/B per layer (mask 0 | 4 | 7), N = 8, then on the whole step
               for m in 0 4 7; do echo "--- DCN_WGRAD_DEEP=$m, N = 8" | tee -a gpurun_out/${tag}_wdab.txt
                 timeout 300 env DCN_WGRAD_DEEP=$m python tools/conv_bench.py --mode f16 --n 8 --x-direct --no-split --kinds wgrad --reps 20 $( [ $m = 7 ] && echo --check ) 2>&1 | grep -v "^net\|Warn\|amdgpu.ids" | cut -c1-60,108-150 | tee -a gpurun_out/${tag}_wdab.txt
               done
               for rep in 1 2 3; do for m in 0 4 7; do
                 timeout 300 env DCN_WGRAD_DEEP=$m python bench.py --steps 20 --warmup 5 --cpu-baseline-steps 0 --no-variants --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abwd wgrad_deep=$m rep=$rep  %.1f images/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/${tag}_wdab.txt
               done; done ;;
    abw)       # same-box A/B of the wgrad tile on the whole step
               for rep in 1 2 3; do for t in 0 128; do
                 timeout 300 env DCN_WGRAD_TILE=$t python bench.py --steps 20 --warmup 5 --cpu-baseline-steps 0 --no-variants --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abw wgrad_tile=$t rep=$rep  %.1f images/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/${tag}_abw.txt
               done; done ;;
    ab)        # same-box A/B, alternating: stream-K completion (inline | kernel) x BN backward reduction (separate 0 | fused 1)
               for rep in 1 2 3; do for v in ${AB_VARIANTS:-inline:0 kernel:0}; do
                 timeout 300 env DCN_GEMM_SK_FIXUP=${v%%:*} DCN_BN_BWD_FUSED=${v##*:} python bench.py --steps 20 --warmup 5 --cpu-baseline-steps 0 --no-variants --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ab sk_fixup=${v%%:*} bn_fused=${v##*:} rep=$rep  %.1f images/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/${tag}_ab.txt
               done; done ;;
    wgab)      # wgrad tile A/B per layer (256-channel / 8-wavefront tile vs 128-channel), N = 8 (config 2) and 16, + numerical check
               for n in 8 16; do for t in 0 128; do echo "--- wgrad tile ${t} (0 = default: 256 on wide layers), N = $n" | tee -a gpurun_out/${tag}_wgab.txt
                 timeout 300 env DCN_WGRAD_TILE=$t python tools/conv_bench.py --mode f16 --n $n --x-direct --no-split --kinds wgrad --only layer --reps 20 $( [ $n = 8 ] && [ $t = 0 ] && echo --check ) 2>&1 | grep -v "^net\|Warn" | cut -c1-200 | tee -a gpurun_out/${tag}_wgab.txt
               done; done ;;
    parity)    timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "conv" > gpurun_out/${tag}_pytest_conv.log 2>&1; tail -5 gpurun_out/${tag}_pytest_conv.log | cut -c1-300 ;;
    *) echo "unknown step $s" ;;
  esac
done
# rocprof databases are large: keep only the summaries
find gpurun_out -name "*.db" -size +20M -delete 2>/dev/null
du -sh gpurun_out 2>/dev/null
echo "=== done $(date +%T)"
