#!/bin/bash
# A/B two builds of the kernel library (build/ab/old.so vs build/ab/new.so) on the same box, interleaved
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
for rep in 1 2 3; do
  for v in old new; do
    cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
    echo "$v $(python bench.py --steps 3000 --warmup 300 --cpu-steps 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_timers_us']['nb_direct']['avg_us'])")"
  done
done
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
