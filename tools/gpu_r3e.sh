#!/bin/bash
# round 2, session 2, visit e: the timing part of visit d again on another box (visit d's box ran latency-bound kernels 20-50 % slower
# than every other box of the round -- k_step_units 26 us against 15, FFT launches 1.5x, the host's CPU baseline 20 % slower too);
# clocks and power state recorded first
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=r3e
( rocm-smi --showclocks --showpower --showperflevel --showtemp 2>&1 | grep -v "^$" | head -40; nproc; grep -m1 "model name" /proc/cpuinfo; uptime ) > gpurun_out/${T}_box_state.txt 2>&1; grep -i "sclk\|power\|perf" gpurun_out/${T}_box_state.txt | head -6 | cut -c1-150
( time timeout 900 python bench.py > gpurun_out/bench_${T}_default.json 2> gpurun_out/bench_${T}_default.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_${T}_default.json | cut -c1-250
( rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk" | head -4 ) >> gpurun_out/${T}_box_state.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${T}_driver.json 2> gpurun_out/bench_${T}_driver.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_${T}_driver.json | cut -c1-250
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T}_dhfr -o trace -- python $R/bench.py --steps 3000 --warmup 300 --cpu-steps 0 --no-scale-workload > $R/gpurun_out/prof_${T}_dhfr.log 2>&1; echo "rocprof dhfr exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T}_w1m -o trace -- python $R/bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-roofline --no-scale-workload > $R/gpurun_out/prof_${T}_w1m.log 2>&1; echo "rocprof w1m exit $?"
cd $R
python tools/rocpd_kernel_stats.py gpurun_out/prof_${T}_dhfr/trace_results.db > gpurun_out/${T}_dhfr_kernel_stats.txt 2>&1; head -8 gpurun_out/${T}_dhfr_kernel_stats.txt | cut -c40-150
python tools/rocpd_kernel_stats.py gpurun_out/prof_${T}_w1m/trace_results.db > gpurun_out/${T}_water1m_kernel_stats.txt 2>&1; head -12 gpurun_out/${T}_water1m_kernel_stats.txt | cut -c40-150
rm -rf gpurun_out/prof_${T}_dhfr gpurun_out/prof_${T}_w1m
for wl in apoa1 water98k water1m; do
  steps=1000; [ $wl = water1m ] && steps=300
  python bench.py --steps $steps --warmup 100 --workload $wl --cpu-steps 0 --no-scale-workload 2>/dev/null | tail -1 > gpurun_out/bench_${T}_$wl.json; cut -c1-200 gpurun_out/bench_${T}_$wl.json
done
python bench.py --steps 3000 --warmup 300 --dt-fs 4 --cpu-steps 0 --no-scale-workload 2>/dev/null | tail -1 > gpurun_out/bench_${T}_dhfr_4fs.json; cut -c1-200 gpurun_out/bench_${T}_dhfr_4fs.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29608 bench.py --gpus 8 --steps 60 --warmup 5 --transport gloo --serialize-ranks --no-scale-workload --no-roofline > gpurun_out/bench_${T}_serialized_n8.json 2> gpurun_out/bench_${T}_serialized_n8.err; echo "serialized N=8 exit $?"
tail -1 gpurun_out/bench_${T}_serialized_n8.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['per_rank_compute_ms_per_step']['ranks'], d['per_rank_compute_ms_per_step']['collectives_per_step'])"
