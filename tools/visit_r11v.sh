#!/bin/bash
# round 5, visit v: where the AMOEBA list builder's time goes (OPENMM_HIP_PL_DEBUG: 4 = per-workgroup trace, +1 = no appends, +2 = no stores)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for dbg in 4 5 6; do
  echo "OPENMM_HIP_PL_DEBUG=$dbg"
  OPENMM_HIP_PL_DEBUG=$dbg timeout 300 python tools/bench_amoeba.py --dhfr --steps 4 --warm 2 2>&1 | grep "pl_build trace" | tail -2 | cut -c1-140
done | tee gpurun_out/r11v_pl_build_split.txt
