#!/bin/bash
# HBM traffic counters (separate passes, as MI355X_MICROARCH.md prescribes) + the part of the GPU test-suite given in PYTEST_ARGS
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ -n "$PYTEST_ARGS" ]; then timeout 900 python -m pytest $PYTEST_ARGS -q -x -m gpu --timeout 600 2>&1 | tail -4; fi
bash tools/gpu_pmc.sh fetch FETCH_SIZE
bash tools/gpu_pmc.sh write WRITE_SIZE
for t in fetch write; do
  f=$(find gpurun_out/pmc_$t -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" nl_find=40 > gpurun_out/pmc_${t}_summary.txt 2>&1 && cat gpurun_out/pmc_${t}_summary.txt
done
