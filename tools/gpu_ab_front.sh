#!/bin/bash
# single stream with / without the fused front launch vs the two-stream default, interleaved on one box
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['roofline']['kernel_timers_us'].items()})"; }
for rep in 1 2; do
  python bench.py --steps 3000 --warmup 300 --cpu-steps 0 2>/dev/null | show "two-stream      "
  python bench.py --steps 3000 --warmup 300 --cpu-steps 0 --props DisablePmeStream=true 2>/dev/null | show "single, fused   "
  OPENMM_HIP_NO_FUSED_FRONT=1 python bench.py --steps 3000 --warmup 300 --cpu-steps 0 --props DisablePmeStream=true 2>/dev/null | show "single, unfused "
done
