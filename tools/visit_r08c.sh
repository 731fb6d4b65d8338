#!/bin/bash
# round 4, visit 8c: amoeba_dhfr knob scan (list skin, predictor points / kind) -- each run: tools/bench_amoeba.py --dhfr, 40 timed steps
cd /root/repo
mkdir -p gpurun_out/r08c
run() { echo "== $*"; env "$@" timeout 300 python tools/bench_amoeba.py --dhfr --steps 40 --warm 10 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ns_per_day'], d['list_builds_per_step'], d['solver_iterations_per_solve'], d['E1'])"; }
{
run A=0
run OPENMM_HIP_AMOEBA_SKIN=0.08
run OPENMM_HIP_AMOEBA_SKIN=0.12
run OPENMM_HIP_AMOEBA_PREDICTOR_POINTS=6
run OPENMM_HIP_AMOEBA_PREDICTOR_POINTS=2
run OPENMM_HIP_AMOEBA_PREDICTOR=poly OPENMM_HIP_AMOEBA_PREDICTOR_POINTS=3
run OPENMM_HIP_AMOEBA_PREDICTOR=poly OPENMM_HIP_AMOEBA_PREDICTOR_POINTS=4
run OPENMM_HIP_AMOEBA_NO_PREDICTOR=1
run OPENMM_HIP_AMOEBA_SKIN=0.08 OPENMM_HIP_AMOEBA_PREDICTOR_POINTS=6
run A=0
} 2>&1 | tee gpurun_out/r08c/amoeba_dhfr_knobs.txt
