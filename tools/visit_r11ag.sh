#!/bin/bash
# round 5, visit ag: AMOEBA -- pl_prepare (tile bounds + displacement check + overflow clear in one launch; this tree) and the number of solver
# iterations enqueued before the host first waits (OPENMM_HIP_AMOEBA_ENQUEUE_SLACK: 1 = one fewer than the last solve took, 0 = as many)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo -n "$1  "; env $2 timeout 300 python tools/bench_amoeba.py $3 --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ns_per_day'], d['solver_iterations_per_solve'])"; }
for rep in 1 2 3; do
  run "dhfr  slack 1" OPENMM_HIP_AMOEBA_ENQUEUE_SLACK=1 --dhfr
  run "dhfr  slack 0" OPENMM_HIP_AMOEBA_ENQUEUE_SLACK=0 --dhfr
done | tee gpurun_out/r11ag_amoeba.txt
for rep in 1 2; do
  run "water slack 1" OPENMM_HIP_AMOEBA_ENQUEUE_SLACK=1 ""
  run "water slack 0" OPENMM_HIP_AMOEBA_ENQUEUE_SLACK=0 ""
done | tee -a gpurun_out/r11ag_amoeba.txt
timeout 1500 python -m pytest tests/test_gpu_platform.py -q -x -k "amoeba" 2>&1 | tail -3 | tee gpurun_out/r11ag_pytest.txt
