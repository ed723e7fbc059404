#!/bin/bash
# Round-4 GPU-box sessions (run through gpurun from the repo root): bash tools/gpu_r4_session.sh <tag> [steps...]
# Everything lands in gpurun_out/<tag>_*; nothing here reads /root/reference.
tag=${1:-r4}; shift
steps=${*:-bench tests4}
mkdir -p gpurun_out
export TMPDIR=/tmp
for s in $steps; do
  echo "=== $s $(date +%T)"
  case $s in
    bench)     timeout 900 python bench.py --steps ${BENCH_STEPS:-20} --warmup ${BENCH_WARMUP:-5} > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err; tail -c 600 gpurun_out/${tag}_bench_default.err
               python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${tag}_bench_default.json').read().strip().splitlines()[-1])
    r=d['roofline']; print('value %.1f images/s  %.2f ms/step  roofline %.3f  hl %.3f  wgrad %.3f  host %.2f' % (d['value'], d['ms_per_step'], r['frac'], r['hl_kernel']['frac'], r['conv_wgrad']['frac'], d['host_enqueue_ms_per_step']))
    e=d['roofline_elementwise']; print('elementwise %.0f GB/s frac %.3f  %.2f ms/step' % (e['achieved'], e['frac'], e['kernel_ms_per_step'])); print({k:(round(v['GBps']),round(v['kernel_ms_per_step'],2)) for k,v in e['passes'].items()}, e['bn_finalize'])
    b=d['breakdown']; print('breakdown: step %.2f  kernel sum %.2f  engine %.2f (%d launches)  loss %.3f  opt %.3f' % (b['ms_per_step'], b['kernel_ms_sum'], b['engine_kernel_ms'], b['engine_launches_per_step'], b['loss_call_ms'], b['optimizer_ms']))
    print({k: round(v,2) for k,v in b['engine_ms_by_category'].items()})
    for k,v in d['variants'].items():
        print(' ', k, round(v['value'],1), round(v['ms_per_step'],2), (v.get('roofline') or {}).get('frac'), {kk: (round(vv,2) if isinstance(vv,float) else vv) for kk,vv in (v.get('breakdown') or {}).items() if kk in ('kernel_ms_sum','engine_kernel_ms','engine_launches_per_step','loss_call_ms','optimizer_ms','host_enqueue_ms_per_step')})
    print('  loss gather', d['roofline_loss_gather']['us_per_call'], d['roofline_loss_gather'].get('at_config3_list_sizes',{}).get('frac_pairs_only'))
    print('  cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
except Exception as ex:
    print('bench line unreadable:', ex)
PY
               ;;
    tests4)    timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q -s -p no:cacheprovider > gpurun_out/${tag}_pytest_r4.log 2>&1; grep -v "^$" gpurun_out/${tag}_pytest_r4.log | tail -25 | cut -c1-700 ;;
    tests)     timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/${tag}_pytest.log 2>&1; tail -30 gpurun_out/${tag}_pytest.log | cut -c1-300 ;;
    smoke)     timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1; tail -3 gpurun_out/${tag}_smoke.log ;;
    prof1)     # kernel stats of a config-1 step (B = 1), pair and two-call pattern
               for sep in "" "--separate-forwards"; do n=$( [ -z "$sep" ] && echo pair || echo separate )
                 (cd /tmp && timeout 600 env DCN_BACKWARD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof1_$n -- python $GRAFT_REPO_ROOT/bench.py --workload config1 $sep --steps 20 --warmup 5 --cpu-baseline-steps 0 --no-variants --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof1_${n}_bench.log 2>&1)
                 python tools/stats_summary.py gpurun_out/${tag}_prof1_$n "config 1 (B = 1), $n, serial schedule: rocprofv3 --kernel-trace --stats -- bench.py --workload config1 $sep --steps 20 --warmup 5" > gpurun_out/${tag}_kernel_stats_config1_$n.txt 2>&1; head -32 gpurun_out/${tag}_kernel_stats_config1_$n.txt | cut -c1-140
                 tail -c 300 gpurun_out/${tag}_prof1_${n}_bench.log | head -c 200; echo
               done ;;
    prof)      (cd /tmp && timeout 900 env DCN_BACKWARD_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline-steps 0 --no-variants --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_bench.log 2>&1); python tools/stats_summary.py gpurun_out/${tag}_prof "config 2, serial schedule: rocprofv3 --kernel-trace --stats -- bench.py --steps 10 --warmup 3 --no-variants" > gpurun_out/${tag}_kernel_stats.txt 2>&1; head -45 gpurun_out/${tag}_kernel_stats.txt | cut -c1-140 ;;
    pmc)       for ctr in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$ctr -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-baseline-steps 0 --no-variants --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$ctr.log 2>&1); done
               python tools/pmc_summary.py gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE gpurun_out/${tag}_hbm_counters.txt gpurun_out/${tag}_hbm_counters.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 4 --warmup 2 --cpu-baseline-steps 0 --no-variants --profile-steps 0" 2>&1 | tail -3; head -16 gpurun_out/${tag}_hbm_counters.txt | cut -c1-150 ;;
    ab)        # in-process A/B: AB_ARGS="--workload config2 --env DCN_BN_REVERSE=0,1,2,3 --reps 3"
               timeout 900 python tools/ab_env.py ${AB_ARGS:---workload config2 --env DCN_BN_REVERSE=0,3 --reps 3} 2>&1 | grep "^ab \|^#" | tee -a gpurun_out/${tag}_ab.txt ;;
    ab2)       timeout 900 python tools/ab_env.py ${AB2_ARGS:---workload config1 --graph 0,1 --separate 0,1 --reps 3 --steps 30 --warmup 8} 2>&1 | grep "^ab \|^#" | tee -a gpurun_out/${tag}_ab2.txt ;;
    ab3)       timeout 900 python tools/ab_env.py ${AB3_ARGS} 2>&1 | grep "^ab \|^#" | tee -a gpurun_out/${tag}_ab3.txt ;;
    convn2)    # per-layer table at N = 2 (config 1: one image pair): fp32-operand kernels, hl32 where eligible, hl32 forced
               for m in "f16:1" "hl:2"; do mode=${m%%:*}; hl=${m##*:}
                 echo "--- mode $mode DCN_GEMM_HL=$hl DCN_WGRAD_HL=$hl, N = 2" | tee -a gpurun_out/${tag}_conv_per_layer_n2.txt
                 timeout 300 env DCN_GEMM_HL=$hl DCN_WGRAD_HL=$hl python tools/conv_bench.py --mode $mode --n 2 --x-direct --reps 20 2>&1 | grep -v "Warn\|amdgpu.ids" | cut -c1-170 | tee -a gpurun_out/${tag}_conv_per_layer_n2.txt
               done ;;
    convn8)    timeout 400 python tools/conv_bench.py --mode hl --n 8 --x-direct --reps 20 --relu-x 2>&1 | grep -v "Warn\|amdgpu.ids" | cut -c1-170 | tee gpurun_out/${tag}_conv_per_layer_n8.txt
               for r in ${HL_ROWS_LIST:-}; do echo "--- DCN_GEMM_HL_ROWS=$r" | tee -a gpurun_out/${tag}_conv_per_layer_n8.txt
                 timeout 300 env DCN_GEMM_HL_ROWS=$r python tools/conv_bench.py --mode hl --n 8 --kinds fwd,dgrad --only "layer4\|layer3" --reps 20 --relu-x 2>&1 | grep -v "Warn\|amdgpu.ids" | cut -c1-170 | tee -a gpurun_out/${tag}_conv_per_layer_n8.txt; done ;;
    tests320)  timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q -p no:cacheprovider -k "320 or identical or profile" > gpurun_out/${tag}_pytest_320.log 2>&1; tail -6 gpurun_out/${tag}_pytest_320.log | cut -c1-300 ;;
    c1sep)     # config 1, two-call pattern, on its own (the in-process variant of the default line measured 35 ms/step, host-bound)
               timeout 300 python bench.py --workload config1 --separate-forwards --no-variants --cpu-baseline-steps 0 --profile-steps 0 --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config1 separate alone: %.1f images/s  %.3f ms/step  host %.2f' % (d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step']))" ;;
    dbg)       timeout 600 python tools/debug/debug_c1sep.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tee gpurun_out/${tag}_debug_c1sep.txt ;;
    dbg5)      timeout 600 python tools/debug/debug_c1sep.py config5 2>&1 | grep -v "Warn\|amdgpu.ids" | tee gpurun_out/${tag}_debug_config5.txt ;;
    testsloss) timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -p no:cacheprovider -k "loss or triplet or step" > gpurun_out/${tag}_pytest_loss.log 2>&1; tail -6 gpurun_out/${tag}_pytest_loss.log | cut -c1-300 ;;
    ab4)       timeout 900 python tools/ab_env.py ${AB4_ARGS} 2>&1 | grep "^ab \|^#" | tee -a gpurun_out/${tag}_ab4.txt ;;
    lossb)     timeout 300 python tools/loss_bench.py --config 3 2>&1 | grep "^loss call" | tee gpurun_out/${tag}_loss_bench.txt
               timeout 300 python tools/loss_bench.py --config 2 --reps 50 2>&1 | grep "^loss call" | tee -a gpurun_out/${tag}_loss_bench.txt
               for m in 00 10; do (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_lossprof_$m -- python $GRAFT_REPO_ROOT/tools/loss_bench.py --config 3 --modes $m > /dev/null 2>&1)
                 python tools/stats_summary.py gpurun_out/${tag}_lossprof_$m "loss call at config-3 sizes, records/prefill = $m" 2>&1 | grep -i "loss\|fill\|kernel |" | head -8 | cut -c1-150 | tee -a gpurun_out/${tag}_loss_bench.txt; done ;;
    tests34)   timeout 1200 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round2.py -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/${tag}_pytest_34.log 2>&1; tail -16 gpurun_out/${tag}_pytest_34.log | cut -c1-200 ;;
    report)    timeout 900 python tools/parity_report.py --json gpurun_out/${tag}_parity_report.json > gpurun_out/${tag}_parity_report.log 2>&1; tail -14 gpurun_out/${tag}_parity_report.log | cut -c1-400 ;;
    sustained) timeout 300 python bench.py --steps 300 --warmup 10 --no-variants --cpu-baseline-steps 0 --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('300 consecutive steps: %.1f images/s  %.3f ms/step  final loss %.5f' % (d['value'], d['ms_per_step'], d['config']['final_loss']))" | tee gpurun_out/${tag}_sustained.txt ;;
    tests4b)   timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q -p no:cacheprovider --durations=5 > gpurun_out/${tag}_pytest_r4b.log 2>&1; tail -14 gpurun_out/${tag}_pytest_r4b.log | cut -c1-200 ;;
    *) echo "unknown step $s" ;;
  esac
done
find gpurun_out -name "*.db" -size +20M -delete 2>/dev/null
du -sh gpurun_out 2>/dev/null
echo "=== done $(date +%T)"
