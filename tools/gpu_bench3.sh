#!/bin/bash
# bench with three profiling settings (overhead of the in-bench HIP-event timers)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in "--no-roofline" "--profile-every 8" "--profile-every 1"; do
  echo "== $mode"
  timeout 600 python bench.py --steps 3000 --warmup 300 --cpu-steps 0 $mode 2> gpurun_out/bench3.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d.get('roofline',{}).get('kernel_timers_us',{}).items()})"
done
