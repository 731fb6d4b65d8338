#!/bin/bash
# round 2, session 2, visit p (the last GPU seconds of the round): list builder without the per-workgroup __threadfence() at its end
# (new) against with it (old), water-1M, two library builds on one box; list tests first
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 60 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "binned or complete or single_image" --timeout 50 2>&1 | grep -E "passed|failed" | tail -1
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
for v in old new old new; do
  cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
  echo "water1m $v $(timeout 60 python bench.py --steps 200 --warmup 50 --prepare-steps 100 --cpu-steps 0 --no-scale-workload --workload water1m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], round(d['roofline']['kernel_timers_us']['nl_update']['avg_us'],1))")"
done 2>&1 | tee gpurun_out/r3p_ab_builder_fence.txt
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
