#!/bin/bash
# round 5, visit bb: AMOEBA -- the tail of the solve (last update, history record, final potentials) enqueued before the host has seen the convergence word (OPENMM_HIP_AMOEBA_NO_SPECULATIVE_TAIL=1: after)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo -n "$1  "; env $2 timeout 300 python tools/bench_amoeba.py $3 --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ns_per_day'], d['solver_iterations_per_solve'], d['E1'])"; }
for rep in 1 2 3; do
  run "dhfr  tail after RT   " OPENMM_HIP_AMOEBA_NO_SPECULATIVE_TAIL=1 --dhfr
  run "dhfr  speculative tail" X=1 --dhfr
done | tee gpurun_out/r11bb_amoeba.txt
for rep in 1 2; do
  run "water tail after RT   " OPENMM_HIP_AMOEBA_NO_SPECULATIVE_TAIL=1 ""
  run "water speculative tail" X=1 ""
done | tee -a gpurun_out/r11bb_amoeba.txt
timeout 600 python tools/diag_amoeba_run_epsilon.py 2>&1 | tail -2 | tee -a gpurun_out/r11bb_amoeba.txt
timeout 1500 python -m pytest tests/test_gpu_platform.py -q -x -k "amoeba" 2>&1 | tail -3 | tee gpurun_out/r11bb_pytest.txt
