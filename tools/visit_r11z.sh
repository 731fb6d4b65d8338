#!/bin/bash
# round 5, visit z: the AMOEBA list builder, lane per owner, float tests, one append loop per 32-slot block
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for dbg in 4 5; do
  echo "OPENMM_HIP_PL_DEBUG=$dbg"
  OPENMM_HIP_PL_DEBUG=$dbg timeout 300 python tools/bench_amoeba.py --dhfr --steps 4 --warm 2 2>&1 | grep "pl_build trace" | tail -2 | cut -c1-140
done | tee gpurun_out/r11z_pl_build_split.txt
for rep in 1 2 3; do timeout 300 python tools/bench_amoeba.py --dhfr --steps 40 2>&1 | tail -1 | cut -c100-420; done | tee gpurun_out/r11z_amoeba_dhfr.txt
timeout 1500 python -m pytest tests/test_gpu_platform.py -q -x -k "amoeba" 2>&1 | tail -3 | tee gpurun_out/r11z_pytest.txt
