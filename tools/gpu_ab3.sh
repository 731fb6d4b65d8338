#!/bin/bash
# A/B over the spatial-sort bin width (rows in the list, builder and pair-kernel time)
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['roofline']['kernel_timers_us'].items()}, 'rows', d['roofline']['rows'], 'rebuilds', d['roofline']['rebuilds'])"; }
for bin in ${BINS:-0.3 0.2 0.15 0.45}; do
  OPENMM_HIP_SORT_BIN=$bin python bench.py --steps 2000 --warmup 300 --cpu-steps 0 2>/dev/null | show "bin $bin"
done
