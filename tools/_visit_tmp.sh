timeout 600 python -m pytest tests/test_gpu_platform.py -m gpu -q -x -s -k "pruned_list" 2>&1 | grep "leg\|passed\|failed" | cut -c1-200
show() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], 'rows', r.get('rows'), r.get('rows_as_built'), 'rebuilds', r.get('rebuilds'), {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items()}, r.get('fp32_issue',{}).get('evals_per_useful_pair'))"; }
export OPENMM_HIP_PRUNE=0; python bench.py --workload water1m --steps 300 --warmup 20 --cpu-steps 0 --no-scale-workload --no-extra-workloads 2>/dev/null | tail -1 | show "w1m prune=0"
export OPENMM_HIP_PRUNE=1
for pad in 0.2 0.3; do for ipad in 0.03 0.05 0.08; do
  export OPENMM_HIP_NL_PADDING=$pad OPENMM_HIP_NL_INNER_PADDING=$ipad
  python bench.py --workload water1m --steps 300 --warmup 20 --cpu-steps 0 --no-scale-workload --no-extra-workloads 2>/dev/null | tail -1 | show "w1m pad=$pad inner=$ipad"
done; done
export OPENMM_HIP_NL_PADDING=0.3 OPENMM_HIP_NL_INNER_PADDING=0.05
python bench.py --workload apoa1 --steps 1000 --warmup 100 --cpu-steps 0 --no-scale-workload --no-extra-workloads 2>/dev/null | tail -1 | show "apoa1 prune pad=0.3 inner=0.05"
