#!/bin/bash
# round 5, visit ar: AMOEBA -- the two dipole sets' reciprocal chains on a side stream each (OPENMM_HIP_AMOEBA_TWO_SIDE_CHAINS=1) against one chain of two-grid launches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo -n "$1  "; env $2 timeout 300 python tools/bench_amoeba.py $3 --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ns_per_day'], d['solver_iterations_per_solve'], d['E1'])"; }
for rep in 1 2 3; do
  run "dhfr  two chains     " OPENMM_HIP_AMOEBA_TWO_SIDE_CHAINS=1 --dhfr
  run "dhfr  one chain      " X=1 --dhfr
done | tee gpurun_out/r11ar_amoeba.txt
for rep in 1 2; do
  run "water two chains     " OPENMM_HIP_AMOEBA_TWO_SIDE_CHAINS=1 ""
  run "water one chain      " X=1 ""
done | tee -a gpurun_out/r11ar_amoeba.txt
