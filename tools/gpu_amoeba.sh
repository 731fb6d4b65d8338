#!/bin/bash
# GPU-box visit for the AMOEBA water workload: tests, timing (tiles / all tiles / direct polarization), kernel statistics.   TAG=... bash tools/gpu_amoeba.sh
cd "$(dirname "$0")/.."
T=${TAG:-amoeba}
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests/test_gpu_platform.py -m gpu -q -x -k "moeba" --timeout 600 > gpurun_out/${T}_pytest_amoeba.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/${T}_pytest_amoeba.log
fi
for v in "" "OPENMM_HIP_AMOEBA_NO_TILES=1"; do env $v timeout 120 python tools/bench_amoeba.py --steps ${STEPS:-10} 2>&1 | tail -1 | tee -a gpurun_out/${T}_amoeba.txt; done
timeout 120 python tools/bench_amoeba.py --steps ${STEPS:-10} --direct 2>&1 | tail -1 | tee -a gpurun_out/${T}_amoeba.txt
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_amo -o trace -- python $R/tools/bench_amoeba.py --steps ${STEPS:-10} > /dev/null 2>&1 )
if [ -f gpurun_out/prof_amo/trace_results.db ]; then python tools/rocpd_kernel_stats.py gpurun_out/prof_amo/trace_results.db > gpurun_out/${T}_amoeba_kernel_stats.txt 2>&1
else f=$(find gpurun_out/prof_amo -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${T}_amoeba_kernel_stats.txt; fi
head -${LINES_SHOWN:-22} gpurun_out/${T}_amoeba_kernel_stats.txt | cut -c1-180
rm -rf gpurun_out/prof_amo
