#!/bin/bash
# round 2, visit y: PME spreading by grid tiles (default above 60 000 atoms) against the brick kernel; tests that go through it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "tiles or pme" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_platform.py tests/test_gpu_multirank.py -m gpu -q -k "water1m or apoa1 or water_1m or four or 1M or 98k" --timeout 1200 2>&1 | tail -3
run() { python bench.py --cpu-steps 0 --no-scale-workload "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['roofline']['kernel_timers_us']; print(d['value'], d['ms_per_step'], 'spread', round(t['pme_spread']['avg_us'],1), 'pairs', round(t['nb_direct']['avg_us'],1), 'interp', round(t['pme_interpolate']['avg_us'],1))"; }
for rep in 1 2; do
  for mode in tiles bricks; do
    [ $mode = bricks ] && export OPENMM_HIP_TILE_SPREAD_MIN_ATOMS=100000000 || unset OPENMM_HIP_TILE_SPREAD_MIN_ATOMS
    echo "apoa1 $mode: $(run --steps 1000 --warmup 100 --workload apoa1)"
    echo "water98k $mode: $(run --steps 1000 --warmup 100 --workload water98k)"
    echo "water1m $mode: $(run --steps 300 --warmup 20 --workload water1m)"
  done
done
unset OPENMM_HIP_TILE_SPREAD_MIN_ATOMS
