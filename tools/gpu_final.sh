#!/bin/bash
# Standard end-of-session visit (tag = $1): whole GPU suite + smoke, PMC passes for the committed kernel sources
# (-> profiles/pmc_pairs_fft.json on the box, so the bench lines below quote a current figure), kernel traces of the DHFR and
# water-1M benches, the default bench line, the driver's command line, the other workloads, the N > 1 rehearsals on one GPU
# (launcher fall-back at N = 2, N = 8 over gloo, 8 ranks serialised), persistent-grid A/B of the pair kernel at 1M atoms
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=${1:-final}
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?"; grep -n "passed\|failed" gpurun_out/pytest_$T.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu_pmc2.sh 2>&1 | grep -A3 "Counter_Name" | head -12 | cut -c1-200
cp gpurun_out/pmc_fetch_summary.txt gpurun_out/${T}_pmc_fetch_summary.txt; cp gpurun_out/pmc_write_summary.txt gpurun_out/${T}_pmc_write_summary.txt
python tools/make_pmc_json.py gpurun_out/pmc_fetch_summary.txt gpurun_out/pmc_write_summary.txt $T > /dev/null 2>&1 && cp profiles/pmc_pairs_fft.json gpurun_out/pmc_pairs_fft.json; echo "pmc json exit $?"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T}_dhfr -o trace -- python $R/bench.py --steps 3000 --warmup 300 --cpu-steps 0 --no-scale-workload > $R/gpurun_out/prof_${T}_dhfr.log 2>&1; echo "rocprof dhfr exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${T}_w1m -o trace -- python $R/bench.py --steps 300 --warmup 20 --workload water1m --cpu-steps 0 --no-roofline --no-scale-workload > $R/gpurun_out/prof_${T}_w1m.log 2>&1; echo "rocprof w1m exit $?"
cd $R
python tools/rocpd_kernel_stats.py gpurun_out/prof_${T}_dhfr/trace_results.db > gpurun_out/${T}_dhfr_kernel_stats.txt 2>&1; head -12 gpurun_out/${T}_dhfr_kernel_stats.txt | cut -c40-150
python tools/rocpd_kernel_stats.py gpurun_out/prof_${T}_w1m/trace_results.db > gpurun_out/${T}_water1m_kernel_stats.txt 2>&1; head -14 gpurun_out/${T}_water1m_kernel_stats.txt | cut -c40-150
rm -rf gpurun_out/prof_${T}_dhfr gpurun_out/prof_${T}_w1m
( time timeout 900 python bench.py > gpurun_out/bench_${T}_default.json 2> gpurun_out/bench_${T}_default.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_${T}_default.json | cut -c1-250
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${T}_driver.json 2> gpurun_out/bench_${T}_driver.err ) 2>&1 | grep real; tail -1 gpurun_out/bench_${T}_driver.json | cut -c1-250
for wl in apoa1 water98k water1m; do
  steps=1000; [ $wl = water1m ] && steps=300
  python bench.py --steps $steps --warmup 100 --workload $wl --cpu-steps 0 --no-scale-workload 2>/dev/null | tail -1 > gpurun_out/bench_${T}_$wl.json; cut -c1-200 gpurun_out/bench_${T}_$wl.json
done
python bench.py --steps 3000 --warmup 300 --dt-fs 4 --cpu-steps 0 --no-scale-workload 2>/dev/null | tail -1 > gpurun_out/bench_${T}_dhfr_4fs.json; cut -c1-200 gpurun_out/bench_${T}_dhfr_4fs.json
# ---- N > 1 rehearsals on this one GPU
t0=$(date +%s)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 5 --attempt-timeout 120 > gpurun_out/bench_${T}_launcher_n2.json 2> gpurun_out/bench_${T}_launcher_n2.err; echo "N=2 exit $? after $(( $(date +%s) - t0 )) s"
grep "launcher" gpurun_out/bench_${T}_launcher_n2.err | head -4 | cut -c1-200; tail -1 gpurun_out/bench_${T}_launcher_n2.json | cut -c1-200
t0=$(date +%s)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 8 --steps 20 --warmup 5 --transport gloo > gpurun_out/bench_${T}_launcher_n8.json 2> gpurun_out/bench_${T}_launcher_n8.err; echo "N=8 exit $? after $(( $(date +%s) - t0 )) s"
tail -1 gpurun_out/bench_${T}_launcher_n8.json | cut -c1-200
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29608 bench.py --gpus 8 --steps 60 --warmup 5 --transport gloo --serialize-ranks --no-scale-workload --no-roofline > gpurun_out/bench_${T}_serialized_n8.json 2> gpurun_out/bench_${T}_serialized_n8.err; echo "serialized N=8 exit $?"
tail -1 gpurun_out/bench_${T}_serialized_n8.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['per_rank_compute_ms_per_step']['ranks'], d['per_rank_compute_ms_per_step']['collectives_per_step'])"
# ---- pair kernel at 1M atoms: one wavefront per chunk (default) against persistent wavefronts walking through the list
for g in 0 2048 4096 8192; do
  ( [ $g != 0 ] && export OPENMM_HIP_DIRECT_GRID=$g
    python bench.py --steps 300 --warmup 100 --workload water1m --cpu-steps 0 --no-scale-workload 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('direct_grid $g', d['value'], d['ms_per_step'], round(d['roofline']['kernel_timers_us']['nb_direct']['avg_us'],1))" )
done 2>&1 | tee gpurun_out/${T}_ab_direct_grid.txt
