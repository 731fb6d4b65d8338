#!/bin/bash
# round 5, visit f: same-box A/B of the five-lane interpolation kernel (default) against the eight-lane kernel, 1M atoms and DHFR
cd "$(dirname "$0")/.."
export TAG=r11f
STEPS=400 BENCH_ARGS="--workload water1m" bash tools/gpu_visit.sh abenv:OPENMM_HIP_INTERPOLATE_LANES=8:-
STEPS=3000 BENCH_ARGS="--no-pmc" bash tools/gpu_visit.sh abenv:OPENMM_HIP_INTERPOLATE_LANES=8:-
STEPS=1000 BENCH_ARGS="--workload apoa1" bash tools/gpu_visit.sh abenv:OPENMM_HIP_INTERPOLATE_LANES=8:-
