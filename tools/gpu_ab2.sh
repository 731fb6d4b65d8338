#!/bin/bash
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['roofline']['kernel_timers_us'].items()}, d['roofline']['rows'], d['roofline']['rebuilds'])"; }
python bench.py --steps 400 --warmup 50 --cpu-steps 0 --profile-every 1 | show "base"
OPENMM_HIP_DEBUG_SPREAD=2 python bench.py --steps 400 --warmup 50 --cpu-steps 0 --profile-every 1 2>/dev/null | show "spread: no flush"
OPENMM_HIP_DEBUG_SPREAD=3 python bench.py --steps 400 --warmup 50 --cpu-steps 0 --profile-every 1 2>/dev/null | show "spread: no accumulate, no flush"
OPENMM_HIP_DEBUG_SPREAD=1 python bench.py --steps 400 --warmup 50 --cpu-steps 0 --profile-every 1 2>/dev/null | show "spread: no accumulate"
OPENMM_HIP_PME_SPREAD_DIRECT=1 python bench.py --steps 400 --warmup 50 --cpu-steps 0 --profile-every 1 2>/dev/null | show "spread: direct atomics"
