#!/bin/bash
# GPU session 2 of round 6: the new tests (order-independent loss backward, bit-reproducible steps, BN kernels), then the bench
# line with the exact loss backward on (default) and off -- same box, back to back.
set -x
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -x --durations=8 > gpurun_out/r6d_new_tests.log 2>&1; tail -14 gpurun_out/r6d_new_tests.log
for rep in 1 2; do
for ex in 1 0; do
  DCN_LOSS_EXACT=$ex timeout 600 python bench.py --no-variants --cpu-baseline-steps 0 > gpurun_out/r6d_bench_exact${ex}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6d_bench_exact${ex}_$rep.json').read().strip().splitlines()[-1])
s=d['summary']; b=d['breakdown']
print('DCN_LOSS_EXACT=${ex} rep $rep: %.1f images/s %.3f ms/step  loss_call %.1f us  finalize %.3f ms (%d launches, avg %.2f us)  elementwise %.3f ms frac %.3f' % (d['value'], d['ms_per_step'], 1e3*b['loss_call_ms'], s['bn_finalize_ms'], s['bn_finalize_launches'], 1e3*s['bn_finalize_ms']/s['bn_finalize_launches'], s['elementwise_ms'], s['frac_elementwise_hbm']))
PY
done; done 2>&1 | tee gpurun_out/r6d_ab_loss_exact.txt
