#!/bin/bash
# round 2, visit c: decomposed path on one GPU (again), list-padding sweep at 1M atoms
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_platform.py -m gpu -q --timeout 600 -s -k "rccl or sharing or water1m" > gpurun_out/pytest_r2c.log 2>&1; echo "pytest exit $?"; grep -h "max-rel-err\|RCCL vs\|forces\|passed\|failed\|Error\|error" gpurun_out/pytest_r2c.log | cut -c1-400 | head -20
for pad in 0.1 0.15 0.2; do
  OPENMM_HIP_NL_PADDING=$pad timeout 400 python bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-scale-workload > gpurun_out/bench_r2c_w1m_pad$pad.json 2> gpurun_out/bench_r2c_w1m_pad$pad.err; echo "pad $pad exit $?"
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_r2c_w1m_pad$pad.json").read().strip().splitlines()[-1])
r=d["roofline"]; t=r["kernel_timers_us"]
print("pad $pad: %.1f ns/day %.3f ms/step rows %d rebuilds %d nb %.0f nl %.0f fft %.0f interp %.0f" % (d["value"], d["ms_per_step"], r["rows"], r["rebuilds"], t["nb_direct"]["avg_us"], t["nl_update"]["avg_us"], t["pme_fft"]["avg_us"], t["pme_interpolate"]["avg_us"]))
PY
done
timeout 400 python bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-scale-workload --decompose > gpurun_out/bench_r2c_w1m_dd1.json 2> gpurun_out/bench_r2c_w1m_dd1.err; echo "w1m dd1 exit $?"; tail -1 gpurun_out/bench_r2c_w1m_dd1.json | cut -c1-700; tail -3 gpurun_out/bench_r2c_w1m_dd1.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2c_w1m_dd1 -o trace -- python $R/bench.py --steps 100 --warmup 20 --workload water1m --cpu-steps 0 --no-roofline --no-scale-workload --decompose > $R/gpurun_out/prof_r2c_w1m_dd1.log 2>&1; echo "rocprof dd1 exit $?"
cd $R
python tools/rocpd_kernel_stats.py gpurun_out/prof_r2c_w1m_dd1/trace_results.db 2>&1 | head -24 | cut -c1-150
