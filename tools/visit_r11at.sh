#!/bin/bash
# round 5, visit at: AMOEBA with the side streams on compute units of their own (OPENMM_HIP_PME_CUS=n: the first n bits of the CU mask for the
# high-priority streams, the rest for the main stream)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { echo -n "$1  "; env $2 timeout 300 python tools/bench_amoeba.py $3 --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ns_per_day'], d['solver_iterations_per_solve'])"; }
for rep in 1 2; do
  run "dhfr  no mask " X=1 --dhfr
  run "dhfr  32 CUs  " OPENMM_HIP_PME_CUS=32 --dhfr
  run "dhfr  64 CUs  " OPENMM_HIP_PME_CUS=64 --dhfr
  run "dhfr  96 CUs  " OPENMM_HIP_PME_CUS=96 --dhfr
done | tee gpurun_out/r11at_amoeba.txt
