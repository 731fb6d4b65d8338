#!/bin/bash
# round 2, visit z: state of the tree at the end of the round -- whole GPU suite, PMC passes for the committed kernel sources, default bench line
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 1200 > gpurun_out/pytest_r2z.log 2>&1; echo "pytest exit $?"; grep -n "passed\|failed" gpurun_out/pytest_r2z.log | tail -2
bash tools/gpu_pmc2.sh 2>&1 | grep -A3 "Counter_Name" | head -12 | cut -c1-200
( time timeout 900 python bench.py > gpurun_out/bench_r2z_default.json 2> gpurun_out/bench_r2z_default.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_r2z_default.json | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
