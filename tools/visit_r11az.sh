#!/bin/bash
# round 5, visit az: AMOEBA -- the list words reach the host among the solver's sums (no copies of their own on the main stream: 33 us in the r11ax timeline); this tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo -n "$1  "; env $2 timeout 300 python tools/bench_amoeba.py $3 --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ns_per_day'], d['solver_iterations_per_solve'], d['E1'])"; }
for rep in 1 2 3; do
  run "dhfr  spine on main  " OPENMM_HIP_AMOEBA_SPINE_MAIN=1 --dhfr
  run "dhfr  spine on side  " X=1 --dhfr
done | tee gpurun_out/r11az_amoeba.txt
for rep in 1 2; do
  run "water spine on main  " OPENMM_HIP_AMOEBA_SPINE_MAIN=1 ""
  run "water spine on side  " X=1 ""
done | tee -a gpurun_out/r11az_amoeba.txt
timeout 600 python tools/diag_amoeba_run_epsilon.py 2>&1 | tail -2 | tee -a gpurun_out/r11az_amoeba.txt
timeout 1500 python -m pytest tests/test_gpu_platform.py -q -x -k "amoeba" 2>&1 | tail -3 | tee gpurun_out/r11az_pytest.txt
