#!/bin/bash
# round 5, visit r: reciprocal-space stream on n compute units of its own (CU masks), the pair kernel's stream on the rest -- two-stream systems
cd "$(dirname "$0")/.."
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('%-26s' % '$1', d['value'], d['ms_per_step'], {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items() if not k.startswith('pairs')})"; }
for wl in apoa1 water1m; do for rep in 1 2; do for n in 0 32 64 96; do
  steps=400; [ $wl = apoa1 ] && steps=1500
  OPENMM_HIP_PME_CUS=$n timeout 600 python bench.py --workload $wl --steps $steps --warmup 100 --cpu-steps 0 --no-extra-workloads --no-scale-workload --no-pmc 2>&1 | tail -1 | show "$wl pme_cus=$n"
done; done; done 2>&1 | tee gpurun_out/r11r_ab_cu_masks.txt
