export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -p no:cacheprovider -k "wgrad" 2>&1 | tail -5 | cut -c1-250 | tee gpurun_out/r3i_pytest_wgrad.log
for mode in f16 hl; do
  echo "--- mode $mode N=8 wgrad" | tee -a gpurun_out/r3i_wgrad_bench.txt
  timeout 300 python tools/conv_bench.py --mode $mode --n 8 --kinds wgrad --only layer --reps 20 --relu-x --x-direct --no-split $( [ $mode = hl ] && echo --check ) 2>&1 | grep "layer3\|layer4\|total" | cut -c1-200 | tee -a gpurun_out/r3i_wgrad_bench.txt
done
