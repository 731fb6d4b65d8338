#!/bin/bash
# round 2, visit p: threshold of the oversized-block list (0.6 / 0.8 / 1.0 of the list cutoff) x builder LDS footprint, 1M atoms
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
for v in big08 big10 big10_lds20; do
  cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
  echo "$v: $(python tools/diag_nl_phases.py 2>&1 | tail -3 | head -2 | tr '\n' ' ')"
done
for rep in 1 2; do
  for v in cols big08 big10 big08_lds20 big10_lds20; do
    cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
    echo "$v $(python bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-scale-workload 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['roofline']['kernel_timers_us']; print(d['ms_per_step'], 'nl', t['nl_update']['avg_us'], 'pairs', t['nb_direct']['avg_us'], 'rebuilds', d['roofline']['rebuilds'])")"
  done
done
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
