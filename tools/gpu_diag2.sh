#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in ${MODES:-staged fused}; do
  if [ $mode = staged ]; then export OPENMM_HIP_DISABLE_FUSED_STEP=1; else unset OPENMM_HIP_DISABLE_FUSED_STEP; fi
  for seed in ${SEEDS:-11 12 13 14 15 16 17 18 19 20 21 22}; do
    TAG=$mode SEED=$seed CHUNK=${CHUNK:-100} TOTAL=${TOTAL:-30000} timeout 300 python tools/diag_dhfr.py run > gpurun_out/diag_${mode}_$seed.log 2>&1
    echo "$mode seed $seed: $(grep NAN gpurun_out/diag_${mode}_$seed.log) $(tail -n 1 gpurun_out/diag_${mode}_$seed.log | cut -c1-60)"
  done
done
