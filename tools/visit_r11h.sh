#!/bin/bash
# round 5, visit h: does the fixed-point grid (integer atomics: DeterministicForces) spread faster at 1M atoms?  + current 1M kernel stats
cd "$(dirname "$0")/.."
R=$(pwd)
for rep in 1 2; do for props in "" "--props DeterministicForces=true"; do
  timeout 600 python bench.py --workload water1m --steps 400 --warmup 100 --cpu-steps 0 --no-extra-workloads --no-scale-workload $props 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('%-40s' % '$props', d['value'], d['ms_per_step'], {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items() if not k.startswith('pairs')})"
done; done 2>&1 | tee gpurun_out/r11h_ab_fixed_point_grid.txt
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_1m -o trace -- python $R/bench.py --workload water1m --steps 300 --warmup 50 --cpu-steps 0 --no-roofline --no-scale-workload --no-extra-workloads --no-pmc --props DisablePmeStream=true > $R/gpurun_out/r11h_water1m.log 2>&1 )
grep "^{" gpurun_out/r11h_water1m.log | cut -c1-200
f=$(find gpurun_out/prof_1m -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_kernel_stats.py "$f" nl_find > gpurun_out/r11h_water1m_single_stream_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_1m
head -14 gpurun_out/r11h_water1m_single_stream_kernel_stats.txt | cut -c1-200; tail -3 gpurun_out/r11h_water1m_single_stream_kernel_stats.txt | cut -c1-300
