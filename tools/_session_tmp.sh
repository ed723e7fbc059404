export TMPDIR=/tmp
mkdir -p gpurun_out
for mode in f16 hl; do
  echo "--- mode $mode N=4 r50" | tee -a gpurun_out/r3h_convbench_r50.txt
  timeout 300 python tools/conv_bench.py --shapes r50 --mode $mode --n 4 --kinds fwd,dgrad --reps 10 --relu-x 2>&1 | grep "r50\|total" | cut -c1-170 | tee -a gpurun_out/r3h_convbench_r50.txt
done
for v in 0 1; do
  timeout 300 env DCN_GEMM_HL=$v python bench.py --workload config5 --steps 6 --warmup 2 --cpu-baseline-steps 0 --no-variants --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5 DCN_GEMM_HL=$v  %.1f images/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/r3h_convbench_r50.txt
done
