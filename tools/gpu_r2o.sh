#!/bin/bash
# round 2, visit o: cell-sorted candidate search by columns (cols) against the per-cell search (base), with and without the smaller
# LDS footprint; list completeness tests on the device
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "cell" 2>&1 | tail -3
cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
for v in cols cols_lds20; do
  cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
  echo "$v: $(python tools/diag_nl_phases.py 2>&1 | tail -1)"
done
for rep in 1 2; do
  for v in base cols cols_lds20; do
    cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
    echo "$v $(python bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-scale-workload 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['roofline']['kernel_timers_us']; print(d['ms_per_step'], 'nl', t['nl_update']['avg_us'], 'pairs', t['nb_direct']['avg_us'], 'rebuilds', d['roofline']['rebuilds'])")"
  done
done
cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so
