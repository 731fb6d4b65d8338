#!/bin/bash
# round 5, visit n: five-lane interpolation kernel at 4 (99 VGPRs), 5 (90), 7 (68) and 8 (64, 4 spilled) wavefronts per SIMD, same box, one stream
cd "$(dirname "$0")/.."
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('%-12s' % '$1', d['value'], d['ms_per_step'], {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items() if k in ('pme_interpolate','pme_spread','nl_update')})"; }
for wl in water1m apoa1 dhfr; do for rep in 1 2; do for v in i4 i5 i6 i8; do
  cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
  steps=400; [ $wl = apoa1 ] && steps=1500; [ $wl = dhfr ] && steps=3000
  timeout 600 python bench.py --workload $wl --steps $steps --warmup 100 --cpu-steps 0 --no-extra-workloads --no-scale-workload --no-pmc --props DisablePmeStream=true 2>/dev/null | tail -1 | show "$wl $v"
done; done; done 2>&1 | tee gpurun_out/r11n_ab_interpolate_occupancy.txt
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
