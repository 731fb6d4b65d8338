#!/bin/bash
# round 5, visit i: i atoms of the pair kernel through LDS broadcasts (-DOMM_I_FROM_LANES=2) against v_readlane (default): parity of the variant, then same-box A/B
cd "$(dirname "$0")/.."
export TAG=r11i
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
cp build/ab/new.so openmm_amd/lib/libopenmm_hip_kernels.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "direct_space or fused or cutoff_edge" 2>&1 | tail -3
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
STEPS=3000 bash tools/gpu_visit.sh ablib:"--no-pmc"
STEPS=400 bash tools/gpu_visit.sh ablib:"--workload water1m"
