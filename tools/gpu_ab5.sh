#!/bin/bash
# pair kernel with and without its force atomics (profiling knob; results of the no-atomics runs are physically wrong)
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], round(d['roofline']['kernel_timers_us']['nb_direct']['avg_us'],1))"; }
for lib in old new; do
  cp build/ab/$lib.so openmm_amd/lib/libopenmm_hip_kernels.so
  for f in 0 1 2 3; do
    OPENMM_HIP_DEBUG_SKIP_ATOMICS=$f python bench.py --steps 300 --warmup 50 --cpu-steps 0 --profile-every 1 --props DisablePmeStream=true 2>/dev/null | show "$lib skip=$f"
  done
done
cp build/ab/old.so openmm_amd/lib/libopenmm_hip_kernels.so
