#!/bin/bash
# round 5, visit ac: AMOEBA -- the vdW force's launch (its list build) after the multipole list build has been enqueued (hook of
# ommhip_amoeba_multipole_forces), against the order before (OPENMM_HIP_AMOEBA_EARLY_FIRST=1); AMOEBA GPU tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2 3; do for v in 1 0; do
  if [ $v = 1 ]; then export OPENMM_HIP_AMOEBA_EARLY_FIRST=1; else unset OPENMM_HIP_AMOEBA_EARLY_FIRST; fi
  echo -n "early_first=$v  "; timeout 300 python tools/bench_amoeba.py --dhfr --steps 40 2>&1 | tail -1 | cut -c140-230
done; done | tee gpurun_out/r11ac_amoeba_hook.txt
unset OPENMM_HIP_AMOEBA_EARLY_FIRST
for rep in 1 2; do for v in 1 0; do
  if [ $v = 1 ]; then export OPENMM_HIP_AMOEBA_EARLY_FIRST=1; else unset OPENMM_HIP_AMOEBA_EARLY_FIRST; fi
  echo -n "water tile, early_first=$v  "; timeout 300 python tools/bench_amoeba.py --steps 40 2>&1 | tail -1 | cut -c100-200
done; done | tee -a gpurun_out/r11ac_amoeba_hook.txt
unset OPENMM_HIP_AMOEBA_EARLY_FIRST
timeout 1500 python -m pytest tests/test_gpu_platform.py -q -x -k "amoeba" 2>&1 | tail -3 | tee gpurun_out/r11ac_pytest.txt
