export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2 3; do for v in 0 1; do
  timeout 300 env DCN_WGRAD_HL=$v python bench.py --steps 20 --warmup 5 --cpu-baseline-steps 0 --no-variants --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abenv DCN_WGRAD_HL=$v rep=$rep  %.1f images/s  %.3f ms/step  loss %.5f' % (d['value'], d['ms_per_step'], d['config']['final_loss']))" | tee -a gpurun_out/r3j_abenv_wgrad_hl.txt
done; done
for v in 0 1; do
  timeout 300 env DCN_WGRAD_HL=$v python bench.py --workload config4 --steps 10 --warmup 3 --cpu-baseline-steps 0 --no-variants --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4 DCN_WGRAD_HL=$v  %.1f images/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/r3j_abenv_wgrad_hl.txt
done
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_round3.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | cut -c1-250 | tee gpurun_out/r3j_pytest.log
