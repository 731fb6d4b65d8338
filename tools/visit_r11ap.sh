#!/bin/bash
# round 5, visit ap: AMOEBA -- the solver's side stream at high priority (OPENMM_HIP_AMOEBA_SIDE_NORMAL=1: as before)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo -n "$1  "; env $2 timeout 300 python tools/bench_amoeba.py $3 --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ns_per_day'], d['solver_iterations_per_solve'], d['E1'])"; }
for rep in 1 2 3; do
  run "dhfr  normal priority" OPENMM_HIP_AMOEBA_SIDE_NORMAL=1 --dhfr
  run "dhfr  high priority  " X=1 --dhfr
done | tee gpurun_out/r11ap_amoeba.txt
for rep in 1 2; do
  run "water normal priority" OPENMM_HIP_AMOEBA_SIDE_NORMAL=1 ""
  run "water high priority  " X=1 ""
done | tee -a gpurun_out/r11ap_amoeba.txt
