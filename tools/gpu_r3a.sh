#!/bin/bash
# round 2, session 2, visit a: GPU suite, then same-box A/B of the two pair-loop changes (polynomial Ewald force, LJ-free block tails)
# on DHFR and on the 1M-atom box, and the driver's 20-step command line (kinetic energy now multi-workgroup)
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_r3a.log 2>&1; echo "pytest exit $?"; grep -n "passed\|failed\|Error" gpurun_out/pytest_r3a.log | tail -4
BENCH_ARGS="--no-scale-workload" bash tools/gpu_ab_env.sh "-" "OPENMM_HIP_NO_EWALD_POLY=1" "OPENMM_HIP_NO_LJ_SPLIT=1" "OPENMM_HIP_NO_EWALD_POLY=1,OPENMM_HIP_NO_LJ_SPLIT=1" 2>&1 | tee gpurun_out/ab_r3a_dhfr.txt | cut -c1-260
STEPS=300 BENCH_ARGS="--no-scale-workload --workload water1m" bash tools/gpu_ab_env.sh "-" "OPENMM_HIP_NO_EWALD_POLY=1,OPENMM_HIP_NO_LJ_SPLIT=1" "OPENMM_HIP_NO_EWALD_POLY=1" 2>&1 | tee gpurun_out/ab_r3a_w1m.txt | cut -c1-260
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r3a_driver.json 2> gpurun_out/bench_r3a_driver.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_r3a_driver.json | cut -c1-300
