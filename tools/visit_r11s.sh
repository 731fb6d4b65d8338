#!/bin/bash
# round 5, visit s: blocks per spreading brick at 92 k atoms (the default switches from one to two at 60 000 atoms) and list padding re-check at 1M
cd "$(dirname "$0")/.."
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('%-34s' % '$1', d['value'], d['ms_per_step'], 'rows', r.get('rows'), 'rebuilds', r.get('rebuilds'), {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items() if not k.startswith('pairs')})"; }
for rep in 1 2; do for g in 1 2; do
  OPENMM_HIP_SPREAD_GROUP=$g timeout 600 python bench.py --workload apoa1 --steps 1500 --warmup 100 --cpu-steps 0 --no-extra-workloads --no-scale-workload --no-pmc 2>&1 | tail -1 | show "apoa1 spread_group=$g"
done; done 2>&1 | tee gpurun_out/r11s_ab_misc.txt
for rep in 1 2; do for pad in 0.12 0.15 0.18; do
  OPENMM_HIP_NL_PADDING=$pad timeout 600 python bench.py --workload water1m --steps 400 --warmup 100 --cpu-steps 0 --no-extra-workloads --no-scale-workload --no-pmc 2>&1 | tail -1 | show "water1m padding=$pad"
done; done 2>&1 | tee -a gpurun_out/r11s_ab_misc.txt
