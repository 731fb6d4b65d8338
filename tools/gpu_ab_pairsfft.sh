#!/bin/bash
# pair kernel travelling with the FFT launches (OPENMM_HIP_PAIRS_WITH_FFT) vs separate launches, interleaved on one box;
# OPENMM_HIP_PAIRS_FFT_SPLIT = list fractions (64ths) at which stages 1 and 2 begin, OPENMM_HIP_PAIRS_FFT_UNIT_ROWS = rows per work unit
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in d['roofline']['kernel_timers_us'].items()})"; }
if [ -n "$RUN_TESTS" ]; then OPENMM_HIP_PAIRS_FFT_UNIT_ROWS=1 OPENMM_HIP_PAIRS_WITH_FFT=1 timeout 600 python -m pytest tests/test_gpu_platform.py -x -q -m gpu 2>&1 | tail -3; fi
for rep in 1 2; do
  timeout 200 python bench.py --steps 3000 --warmup 300 --cpu-steps 0 2>/dev/null | show "separate        "
  for cfg in ${CFGS:-2:21,43 1:21,43 1:16,40 1:24,46}; do
    OPENMM_HIP_PAIRS_FFT_UNIT_ROWS=${cfg%%:*} OPENMM_HIP_PAIRS_FFT_SPLIT=${cfg##*:} OPENMM_HIP_PAIRS_WITH_FFT=1 timeout 200 python bench.py --steps 3000 --warmup 300 --cpu-steps 0 2>/dev/null | show "pairs+fft $cfg "
  done
done
