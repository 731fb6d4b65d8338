#!/bin/bash
# A/B two builds of the kernel library on a given workload (fewer steps for large systems)
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
for v in old new old new; do
  cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
  echo "$v $(python bench.py --cpu-steps 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k: round(v['avg_us'],1) for k,v in d['roofline']['kernel_timers_us'].items()})")"
done
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
