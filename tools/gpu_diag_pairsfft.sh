#!/bin/bash
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 OPENMM_HIP_PAIRS_WITH_FFT=1
for dbg in ${DBGS:-0 1 2 3}; do
  TAG="debug=$dbg" OPENMM_HIP_PAIRS_FFT_DEBUG=$dbg timeout 120 python tools/diag_pairsfft.py 2>&1 | tail -1
done
