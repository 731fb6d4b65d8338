#!/bin/bash
# round 2, session 2, visit n: CM-momentum tail of the fused integration kernel -- block partials handed over with returning atomics
# (new) against plain stores + __threadfence() (old); two builds of the kernel library interleaved on one box, DHFR and water-1M;
# then the GPU tests that exercise the CM remover and the decomposed runs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
for wl in dhfr water1m; do
  steps=3000; [ $wl = water1m ] && steps=300
  for rep in 1 2 3; do
    for v in old new; do
      [ $wl = water1m ] && [ $rep = 3 ] && continue
      cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
      echo "$wl $v $(timeout 300 python bench.py --steps $steps --warmup 300 --cpu-steps 0 --no-scale-workload --no-roofline --workload $wl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
    done
  done
done 2>&1 | tee gpurun_out/r3n_ab_cm_tail.txt
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
timeout 600 python -m pytest tests/test_gpu_platform.py tests/test_gpu_multirank.py -m gpu -q -x -k "CMMotion or fused_step or dhfr_runs or multirank or rank or invariants" --timeout 400 2>&1 | tail -2
