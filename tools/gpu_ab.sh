#!/bin/bash
# A/B bench variants on the GPU; prints value, ms/step and the kernel timers
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['roofline']['kernel_timers_us'].items()}, d['roofline']['rows'], d['roofline']['rebuilds'])"; }
python bench.py --steps 1500 --warmup 150 --cpu-steps 0 | show "A pme-stream"
python bench.py --steps 1500 --warmup 150 --cpu-steps 0 --props DisablePmeStream=true | show "B single-stream"
OPENMM_HIP_DEBUG_SKIP_ATOMICS=1 python bench.py --steps 300 --warmup 50 --cpu-steps 0 --props DisablePmeStream=true 2>/dev/null | show "C no-j-atomics" || echo "C failed"
OPENMM_HIP_DEBUG_SKIP_ATOMICS=3 python bench.py --steps 300 --warmup 50 --cpu-steps 0 --props DisablePmeStream=true 2>/dev/null | show "D no-atomics" || echo "D failed"
OPENMM_HIP_NL_PADDING=0.15 python bench.py --steps 1500 --warmup 150 --cpu-steps 0 --props DisablePmeStream=true | show "E pad0.15"
OPENMM_HIP_NL_PADDING=0.2 python bench.py --steps 1500 --warmup 150 --cpu-steps 0 --props DisablePmeStream=true | show "F pad0.20"
