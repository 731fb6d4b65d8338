#!/bin/bash
# round 2, visit r: whole GPU suite on the new builder defaults; list padding and cell-mode threshold re-tuned; DHFR regression check
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 1200 > gpurun_out/pytest_r2r.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_r2r.log
run() { python bench.py --cpu-steps 0 --no-scale-workload "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['roofline']['kernel_timers_us']; print(d['value'], d['ms_per_step'], 'nl', round(t['nl_update']['avg_us'],1), 'pairs', round(t['nb_direct']['avg_us'],1), 'rebuilds', d['roofline']['rebuilds'])"; }
for rep in 1 2; do
  for pad in 0.2 0.175 0.15; do echo "water1m padding $pad: $(OPENMM_HIP_NL_PADDING=$pad run --steps 300 --warmup 20 --workload water1m)"; done
done
for rep in 1 2; do
  for cm in 16384 1000; do
    echo "apoa1 cells>=$cm: $(OPENMM_HIP_NL_CELL_MIN_BLOCKS=$cm run --steps 1000 --warmup 100 --workload apoa1)"
    echo "water98k cells>=$cm: $(OPENMM_HIP_NL_CELL_MIN_BLOCKS=$cm run --steps 1000 --warmup 100 --workload water98k)"
  done
done
for rep in 1 2; do echo "dhfr: $(run --steps 3000 --warmup 300)"; done
