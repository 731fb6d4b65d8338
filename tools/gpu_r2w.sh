#!/bin/bash
# round 2, visit w: whole GPU suite after the FFT / builder / binning changes; water-1M and DHFR
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 1200 > gpurun_out/pytest_r2w.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_r2w.log
run() { python bench.py --cpu-steps 0 --no-scale-workload "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['roofline']['kernel_timers_us']; print(d['value'], d['ms_per_step'], 'nl', round(t['nl_update']['avg_us'],1), 'pairs', round(t['nb_direct']['avg_us'],1), 'rebuilds', d['roofline']['rebuilds'])"; }
for rep in 1 2; do echo "water1m: $(run --steps 300 --warmup 20 --workload water1m)"; done
for rep in 1 2; do echo "dhfr: $(run --steps 3000 --warmup 300)"; done
