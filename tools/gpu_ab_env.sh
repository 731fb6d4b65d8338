#!/bin/bash
# same-box interleaved comparison of environment-knob settings: gpu_ab_env.sh "VAR=a" "VAR=b" ...   (each argument = one configuration,
# several assignments separated by commas; "-" = defaults); BENCH_ARGS for extra bench arguments
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s' % '$1', d['value'], d['ms_per_step'], 'rows', r['rows'], 'rebuilds', r['rebuilds'], {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r['kernel_timers_us'].items()})"; }
for rep in 1 2; do
  for cfg in "$@"; do
    ( if [ "$cfg" != "-" ]; then for kv in ${cfg//,/ }; do export "$kv"; done; fi
      timeout 300 python bench.py --steps ${STEPS:-3000} --warmup 300 --cpu-steps 0 $BENCH_ARGS 2>/dev/null | show "$cfg" )
  done
done
