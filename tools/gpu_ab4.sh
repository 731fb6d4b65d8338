#!/bin/bash
# A/B over the reorder interval
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['roofline']['kernel_timers_us'].items()}, 'rows', d['roofline']['rows'], 'rebuilds', d['roofline']['rebuilds'])"; }
for iv in ${IVS:-500 1000 2000 4000}; do
  OPENMM_HIP_REORDER_INTERVAL=$iv python bench.py --steps 8000 --warmup 300 --cpu-steps 0 2>/dev/null | show "interval $iv"
done
