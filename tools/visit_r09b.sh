#!/bin/bash
# round 4, visit 9b: where the half-row build loses -- kernel structure alone (k1n0: classes all "both"), builder order alone (k0n1), both (new), neither (old)
cd /root/repo
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('%-6s' % '$1', d['value'], d['ms_per_step'], 'rows', r.get('rows'), 'half', r.get('half_rows'), {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items() if k in ('nb_direct','nl_update','pairs_fft_stage0')})"; }
for args in "" "--workload water1m --steps 300 --warmup 100"; do
  echo "==== $args"
  for rep in 1 2; do for v in old k1n0 k0n1 new; do
    cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
    timeout 600 python bench.py --steps 3000 --warmup 300 --cpu-steps 0 --no-extra-workloads --no-scale-workload $args 2>/dev/null | tail -1 | show $v
  done; done
done 2>&1 | tee gpurun_out/r09b_ab_half_rows_parts.txt
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
