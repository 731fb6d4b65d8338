#!/bin/bash
# round 2, visit x (final numbers): PMC passes for the fused pair/FFT launches, kernel traces of the DHFR and water-1M benches, the
# default bench line, the driver's command line, apoa1 / water98k / water-1M lines
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_pmc2.sh 2>&1 | tail -24 | cut -c1-220
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2x_dhfr -o trace -- python $R/bench.py --steps 3000 --warmup 300 --cpu-steps 0 --no-scale-workload > $R/gpurun_out/prof_r2x_dhfr.log 2>&1; echo "rocprof dhfr exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2x_w1m -o trace -- python $R/bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-roofline --no-scale-workload > $R/gpurun_out/prof_r2x_w1m.log 2>&1; echo "rocprof w1m exit $?"
cd $R
python tools/rocpd_kernel_stats.py gpurun_out/prof_r2x_dhfr/trace_results.db > gpurun_out/r02x_dhfr_kernel_stats.txt 2>&1; head -12 gpurun_out/r02x_dhfr_kernel_stats.txt | cut -c40-150
python tools/rocpd_kernel_stats.py gpurun_out/prof_r2x_w1m/trace_results.db > gpurun_out/r02x_water1m_kernel_stats.txt 2>&1; head -14 gpurun_out/r02x_water1m_kernel_stats.txt | cut -c40-150
rm -rf gpurun_out/prof_r2x_dhfr gpurun_out/prof_r2x_w1m
( time timeout 900 python bench.py > gpurun_out/bench_r2x_default.json 2> gpurun_out/bench_r2x_default.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_r2x_default.json | cut -c1-250
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r2x_driver.json 2> gpurun_out/bench_r2x_driver.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_r2x_driver.json | cut -c1-250
for wl in apoa1 water98k water1m; do
  steps=1000; [ $wl = water1m ] && steps=300
  python bench.py --steps $steps --warmup 100 --workload $wl --cpu-steps 0 --no-scale-workload 2>/dev/null | tail -1 > gpurun_out/bench_r2x_$wl.json; cut -c1-200 gpurun_out/bench_r2x_$wl.json
done
python bench.py --steps 3000 --warmup 300 --dt-fs 4 --cpu-steps 0 --no-scale-workload 2>/dev/null | tail -1 > gpurun_out/bench_r2x_dhfr_4fs.json; cut -c1-200 gpurun_out/bench_r2x_dhfr_4fs.json
