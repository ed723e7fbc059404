#!/bin/bash
# Round-end GPU session (run through gpurun from the repo root): full -m gpu suite, smoke, the default bench line, kernel
# stats, HBM counters of the gather-GEMM kernels, SQ counters of the hl32 kernels.  bash tools/gpu_final_session.sh <tag> [steps...]
tag=${1:-r3}; shift
steps=${*:-tests smoke bench prof pmc sq}
mkdir -p gpurun_out
export TMPDIR=/tmp
for s in $steps; do
  echo "=== $s $(date +%T)"
  case $s in
    tests) timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${tag}_pytest.log 2>&1; tail -6 gpurun_out/${tag}_pytest.log | cut -c1-300 ;;
    smoke) timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1; tail -3 gpurun_out/${tag}_smoke.log ;;
    bench) timeout 900 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err; tail -c 400 gpurun_out/${tag}_bench_default.err
           python - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print('value %.1f images/s  %.2f ms/step  roofline %.3f  hl %.3f  wgrad %.3f' % (d['value'], d['ms_per_step'], r['frac'], r['hl_kernel']['frac'], r['conv_wgrad']['frac']))
for k,v in d['variants'].items(): print(' ', k, round(v['value'],1), round(v['ms_per_step'],2), (v.get('roofline') or {}).get('frac'))
print('  loss gather', d['roofline_loss_gather']['us_per_call'], d['roofline_loss_gather'].get('at_config3_list_sizes',{}).get('frac_pairs_only'))
PY
           ;;
    graph) timeout 600 python bench.py --hip-graph --no-variants --cpu-baseline-steps 0 --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hip-graph: %.1f images/s  %.3f ms/step  (%s)' % (d['value'], d['ms_per_step'], d['config']['hip_graph']))" ;;
    prof)  (cd /tmp && timeout 900 env DCN_BACKWARD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline-steps 0 --no-variants --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_bench.log 2>&1); python tools/stats_summary.py gpurun_out/${tag}_prof > gpurun_out/${tag}_kernel_stats.txt 2>&1; head -12 gpurun_out/${tag}_kernel_stats.txt | cut -c1-150 ;;
    pmc)   for ctr in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$ctr -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-baseline-steps 0 --no-variants --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$ctr.log 2>&1); done
           python tools/pmc_summary.py gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE gpurun_out/${tag}_hbm_counters.txt gpurun_out/${tag}_hbm_counters.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 4 --warmup 2 --cpu-baseline-steps 0 --no-variants --profile-steps 0" 2>&1 | tail -3; head -14 gpurun_out/${tag}_hbm_counters.txt | cut -c1-150 ;;
    sq)    i=0; for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do i=$((i+1))
             (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_sq_$i -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --mode hl --n 8 --kinds fwd,dgrad,wgrad --x-direct --no-split --only "3x3 d" --reps 5 --relu-x > $GRAFT_REPO_ROOT/gpurun_out/${tag}_sq_$i.log 2>&1)
           done
           python tools/sq_summary.py gpurun_out/${tag}_hl_sq_counters.txt "rocprofv3 --kernel-trace --pmc <two passes> -- python tools/conv_bench.py --mode hl --n 8 --kinds fwd,dgrad,wgrad --x-direct --no-split --only '3x3 d' --reps 5 --relu-x" gpurun_out/${tag}_sq_1 gpurun_out/${tag}_sq_2; cat gpurun_out/${tag}_hl_sq_counters.txt | cut -c1-120 ;;
    *) echo "unknown step $s" ;;
  esac
done
find gpurun_out -name "*.db" -size +20M -delete 2>/dev/null
echo "=== done $(date +%T)"
