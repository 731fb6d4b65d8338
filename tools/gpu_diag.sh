#!/bin/bash
# long-run stability of the DHFR-like workload: several thermostat seeds, 10 000 steps each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for seed in ${SEEDS:-1 2 3 4 5 6}; do
  SEED=$seed CHUNK=500 TOTAL=${TOTAL:-10000} timeout 300 python tools/diag_dhfr.py run > gpurun_out/diag_lds_$seed.log 2>&1
  echo "seed $seed: $(grep -c step gpurun_out/diag_lds_$seed.log) chunks, $(grep NAN gpurun_out/diag_lds_$seed.log) $(tail -n 1 gpurun_out/diag_lds_$seed.log)"
done
