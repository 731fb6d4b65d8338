#!/bin/bash
# round 2, visit s: line-pass FFT with the next tile's reads in flight during the passes of the current one
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "fft or pme" 2>&1 | tail -2
run() { python bench.py --cpu-steps 0 --no-scale-workload "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d.get('roofline_fft',{}); print(d['value'], d['ms_per_step'], 'fft', f.get('grid'), f.get('avg_us'), 'frac', f.get('frac'))"; }
for rep in 1 2; do
  echo "water1m: $(run --steps 300 --warmup 20 --workload water1m)"
  echo "apoa1: $(run --steps 1000 --warmup 100 --workload apoa1)"
  echo "water98k: $(run --steps 1000 --warmup 100 --workload water98k)"
done
