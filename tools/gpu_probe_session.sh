#!/bin/bash
# GPU-box session for the stand-alone GEMM probes (run through gpurun from the repo root):
#   bash tools/gpu_probe_session.sh <tag> [run] [sq]
# run: all variants of tools/hl_probe2.bin (prebuilt here: hipcc --offload-arch=gfx950 -O3 -o tools/hl_probe2.bin tools/hl_gemm_probe2.hip)
# sq : SQ counters (two --pmc passes) of the variants selected by PROBE_SQ_MASKS (default "1 4": plain loop, ping-pong + setprio)
tag=${1:-r3}; shift
steps=${*:-run sq}
mkdir -p gpurun_out
export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/tools/hl_probe2.bin
[ -x $BIN ] || { hipcc --offload-arch=gfx950 -O3 -o /tmp/hl_probe2 tools/hl_gemm_probe2.hip && BIN=/tmp/hl_probe2; }
for s in $steps; do
  echo "=== $s $(date +%T)"
  case $s in
    run) for shape in ${PROBE_SHAPES:-"65536 512 4608" "38400 512 4608" "38400 256 2304"}; do
           timeout 300 $BIN $shape ${PROBE_MASK:-0xffffffff} 10 2 2>&1 | tee -a gpurun_out/${tag}_probe2.txt
         done ;;
    sq)  for m in ${PROBE_SQ_MASKS:-1 4}; do i=0
           for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do i=$((i+1))
             (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_probe_sq_${m}_$i -- $BIN 65536 512 4608 $m 4 1 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_probe_sq_${m}_$i.log 2>&1)
           done
           python tools/sq_summary.py gpurun_out/${tag}_probe_sq_counters_mask$m.txt "rocprofv3 --kernel-trace --pmc <two passes> -- tools/hl_probe2.bin 65536 512 4608 $m 4 1" gpurun_out/${tag}_probe_sq_${m}_1 gpurun_out/${tag}_probe_sq_${m}_2; cat gpurun_out/${tag}_probe_sq_counters_mask$m.txt
         done ;;
    *) echo "unknown step $s" ;;
  esac
done
find gpurun_out -name "*.db" -size +20M -delete 2>/dev/null
echo "=== done $(date +%T)"
