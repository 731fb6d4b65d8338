#!/bin/bash
# round 2, visit f: whole GPU suite, PMC passes (HBM traffic of the fused pair/FFT launches), kernel trace of the DHFR bench,
# the driver's own command line
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/pytest_r2f.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/pytest_r2f.log | cut -c1-300
bash tools/gpu_pmc2.sh 2>&1 | tail -30 | cut -c1-220
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2f_dhfr -o trace -- python $R/bench.py --steps 3000 --warmup 300 --cpu-steps 0 --no-scale-workload > $R/gpurun_out/prof_r2f_dhfr.log 2>&1; echo "rocprof exit $?"
cd $R
python tools/rocpd_kernel_stats.py gpurun_out/prof_r2f_dhfr/trace_results.db 2>&1 | head -14 | cut -c1-150
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r2f_driver.json 2> gpurun_out/bench_r2f_driver.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_r2f_driver.json | cut -c1-300; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/bench_r2f_driver.json").read().splitlines() if l.startswith("{")][-1])
print({k: d[k] for k in ("value","ms_per_step","scaling")}, d.get("scale_workload"), d.get("roofline_fft"))
PY
