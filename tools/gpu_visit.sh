#!/bin/bash
# ONE script for every kind of GPU-box visit (run through `gpurun -- 'bash tools/gpu_visit.sh <steps...>'` from the repository
# root).  A visit is a sequence of steps, each `name[:arg[:arg...]]`; everything is written under gpurun_out/ with the tag given
# by TAG (default "visit"); summaries worth keeping are copied to profiles/ by hand afterwards.
#
#   tests[:pytest-args]          pytest -m gpu (whole suite, or e.g. tests:"tests/test_gpu_kernels.py -k edge"), then smoke()
#   bench:name[:bench-args]      one bench line -> gpurun_out/<TAG>_bench_<name>.json   (bench:driver:"--steps 20 --warmup 5")
#   trace:name[:bench-args]      rocprofv3 --kernel-trace --stats of a bench command -> <TAG>_<name>_kernel_stats.txt
#   pmc:name:"CTR ..."[:bench-args]   one rocprofv3 --pmc pass (own run, kernel trace only, as gpurun requires) -> <TAG>_pmc_<name>.txt
#   hbm                          FETCH_SIZE and WRITE_SIZE passes of the default bench + profiles/pmc_pairs_fft.json on the box
#   sq:name[:bench-args]         the two SQ passes that say what a kernel's waves do (busy / parked on s_waitcnt or a barrier / issue-stalled)
#   ablib[:bench-args]           same-box interleaved A/B of build/ab/old.so against build/ab/new.so (3 rounds; prints ns/day and timers)
#   abenv:"A=1,B=2":"-"[:...]    same-box interleaved A/B of environment-knob settings ("-" = defaults); BENCH_ARGS for bench arguments
#   ranks:N[:bench-args]         N ranks of the decomposed run on the visible GPUs (RCCL when there are N GPUs, else the launcher falls back)
#   serial:N                     N ranks serialised on one GPU over the host-staged transport: per-rank compute per step
#   serialtrace:N                the same under rocprofv3 --kernel-trace --stats (per-rank kernel statistics)
#   serialtimeline:N[:count]     the same under rocprofv3: timeline of the busiest rank's last dispatches + its kernel statistics
#   alone:N[:steps]              each rank of an N-rank decomposition alone on this GPU, no communication, streams overlapped: per-rank ms per step
#   sh:"command"                 anything else
cd "$(dirname "$0")/.."
R=$(pwd); T=${TAG:-visit}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0

kstats() {   # rocprofv3 output directory, output file
  if [ -f "$1/trace_results.db" ]; then python tools/rocpd_kernel_stats.py "$1/trace_results.db" > "$2" 2>&1
  else f=$(find "$1" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$2"; fi
  head -16 "$2" | cut -c1-170
}
pmc_pass() { # tag, counters, bench args
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$1 -o pmc -- \
      python $R/bench.py --steps 60 --warmup 10 --cpu-steps 0 --no-roofline --no-scale-workload --no-extra-workloads $3 > $R/gpurun_out/pmc_$1.log 2>&1; echo "rocprof pmc $1 exit $?" )
  f=$(find gpurun_out/pmc_$1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" nl_find=40 > gpurun_out/${T}_pmc_$1.txt 2>&1 && grep -v "^Scratch\|^LDS_Block\|^Accum" gpurun_out/${T}_pmc_$1.txt | cut -c1-200
  rm -rf gpurun_out/pmc_$1
}
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('%-28s' % '$1', d['value'], d['ms_per_step'], 'rows', r.get('rows'), {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items()})"; }

for step in "$@"; do
  IFS=':' read -r kind a1 a2 a3 <<< "$step"
  echo "==== $step"
  case $kind in
    tests)
      timeout 2400 python -m pytest ${a1:-tests} -m gpu -q -x --timeout 900 > gpurun_out/${T}_pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/${T}_pytest.log
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)
      ( time timeout 1500 python bench.py $a2 > gpurun_out/${T}_bench_$a1.json 2> gpurun_out/${T}_bench_$a1.err ) 2>&1 | grep real; tail -1 gpurun_out/${T}_bench_$a1.json | cut -c1-400 ;;
    trace)
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$a1 -o trace -- python $R/bench.py --cpu-steps 0 --no-extra-workloads $a2 > $R/gpurun_out/prof_$a1.log 2>&1; echo "rocprof exit $?" )
      kstats gpurun_out/prof_$a1 gpurun_out/${T}_${a1}_kernel_stats.txt; rm -rf gpurun_out/prof_$a1 ;;
    pmc) pmc_pass "$a1" "$a2" "$a3" ;;
    hbm)
      pmc_pass fetch FETCH_SIZE ""; pmc_pass write WRITE_SIZE ""
      python tools/make_pmc_json.py gpurun_out/${T}_pmc_fetch.txt gpurun_out/${T}_pmc_write.txt $T > /dev/null 2>&1 && cp profiles/pmc_pairs_fft.json gpurun_out/pmc_pairs_fft.json; echo "pmc json exit $?" ;;
    sq)
      pmc_pass ${a1}_1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "$a2"
      pmc_pass ${a1}_2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "$a2" ;;
    ablib)
      cp openmm_amd/lib/libopenmm_hip_kernels.so /tmp/keep.so
      for rep in 1 2 3; do for v in old new; do
        cp build/ab/$v.so openmm_amd/lib/libopenmm_hip_kernels.so
        timeout ${RUN_TIMEOUT:-600} python bench.py --steps ${STEPS:-3000} --warmup 300 --cpu-steps 0 --no-extra-workloads --no-scale-workload $a1 2>/dev/null | tail -1 | show $v
      done; done 2>&1 | tee -a gpurun_out/${T}_ablib.txt
      cp /tmp/keep.so openmm_amd/lib/libopenmm_hip_kernels.so ;;
    abenv)
      for rep in 1 2; do for cfg in "$a1" "$a2" $a3; do
        ( if [ "$cfg" != "-" ]; then for kv in ${cfg//,/ }; do export "$kv"; done; fi
          timeout ${RUN_TIMEOUT:-600} python bench.py --steps ${STEPS:-3000} --warmup 300 --cpu-steps 0 --no-extra-workloads --no-scale-workload $BENCH_ARGS 2>/dev/null | tail -1 | show "$cfg" )
      done; done 2>&1 | tee -a gpurun_out/${T}_abenv.txt ;;
    ranks)
      t0=$(date +%s)
      timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $a1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) bench.py --gpus $a1 --steps 20 --warmup 5 --attempt-timeout 150 $a2 \
        > gpurun_out/${T}_bench_n$a1.json 2> gpurun_out/${T}_bench_n$a1.err; echo "N=$a1 exit $? after $(( $(date +%s) - t0 )) s"
      grep "launcher" gpurun_out/${T}_bench_n$a1.err | head -4 | cut -c1-200; tail -1 gpurun_out/${T}_bench_n$a1.json | cut -c1-400 ;;
    serial)
      timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $a1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) bench.py --gpus $a1 --steps 60 --warmup 5 --transport gloo --serialize-ranks --no-scale-workload --no-roofline \
        > gpurun_out/${T}_serialized_n$a1.json 2> gpurun_out/${T}_serialized_n$a1.err; echo "serialized N=$a1 exit $?"
      tail -1 gpurun_out/${T}_serialized_n$a1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['per_rank_compute_ms_per_step']['ranks'], d['per_rank_compute_ms_per_step']['collectives_per_step'])" ;;
    serialtrace)     # the same under rocprofv3: kernel statistics of every rank's process (one file per rank; the busiest is shown)
      ( cd /tmp && export TMPDIR=/tmp BENCH_NO_HARD_EXIT=1 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sn$a1 -o trace -- python -m torch.distributed.run --nnodes=1 --nproc-per-node $a1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) $R/bench.py --gpus $a1 --steps 60 --warmup 5 --transport gloo --serialize-ranks --no-scale-workload --no-roofline \
        > $R/gpurun_out/${T}_serialtrace_n$a1.json 2> $R/gpurun_out/${T}_serialtrace_n$a1.err; echo "rocprof serialized N=$a1 exit $?" )
      k=0; for f in $(find gpurun_out/prof_sn$a1 -name "*kernel_stats.csv" | xargs ls -S); do k=$((k+1)); cp $f gpurun_out/${T}_serial_n${a1}_kernel_stats_$k.csv; done
      head -22 gpurun_out/${T}_serial_n${a1}_kernel_stats_1.csv | cut -c1-200; rm -rf gpurun_out/prof_sn$a1 ;;
    serialtimeline)  # the same under rocprofv3: the last dispatches of the busiest rank's process as a timeline (what runs beside what, per queue)
      ( cd /tmp && export TMPDIR=/tmp BENCH_NO_HARD_EXIT=1 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_st$a1 -o trace -- python -m torch.distributed.run --nnodes=1 --nproc-per-node $a1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) $R/bench.py --gpus $a1 --steps 40 --warmup 5 --transport gloo --serialize-ranks --no-scale-workload --no-roofline \
        > $R/gpurun_out/${T}_serialtimeline_n$a1.json 2> $R/gpurun_out/${T}_serialtimeline_n$a1.err; echo "rocprof serialized N=$a1 exit $?" )
      f=$(find gpurun_out/prof_st$a1 -name "*kernel_trace.csv" | xargs -r ls -S | head -1)
      if [ -n "$f" ]; then python tools/rocpd_timeline.py $f ${a2:-160} > gpurun_out/${T}_serial_n${a1}_timeline.txt 2>&1; cp ${f%kernel_trace.csv}kernel_stats.csv gpurun_out/${T}_serial_n${a1}_kernel_stats.csv; else find gpurun_out/prof_st$a1 | head; fi
      tail -${a2:-160} gpurun_out/${T}_serial_n${a1}_timeline.txt | cut -c1-150; rm -rf gpurun_out/prof_st$a1 ;;
    alone)           # every rank of an N-rank decomposition of the 1M-atom box ALONE on this GPU (collectives that cost nothing): bench.py --rank-alone
      timeout 900 python bench.py --rank-alone $a1 --steps ${a2:-400} --warmup 40 > gpurun_out/${T}_rank_alone_n$a1.json 2> gpurun_out/${T}_rank_alone_n$a1.err; echo "alone N=$a1 exit $?"
      tail -1 gpurun_out/${T}_rank_alone_n$a1.json | cut -c1-200 ;;
    sh) bash -c "$a1" ;;
    *) echo "unknown step $kind" ;;
  esac
done
