timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -2
show() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], 'rows', r.get('rows'), r.get('rows_as_built'), 'rebuilds', r.get('rebuilds'), {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items()}, r.get('fp32_issue',{}).get('evals_per_useful_pair'))"; }
for v in 1 0; do
  export OPENMM_HIP_PRUNE=$v
  python bench.py --steps 3000 --warmup 300 --cpu-steps 0 --no-scale-workload --no-extra-workloads 2>/dev/null | tail -1 | show "dhfr prune=$v"
done
for pad in 0.2 0.25 0.3; do for v in 1 0; do
  export OPENMM_HIP_PRUNE=$v OPENMM_HIP_NL_PADDING=$pad
  python bench.py --workload water1m --steps 300 --warmup 20 --cpu-steps 0 --no-scale-workload --no-extra-workloads 2>/dev/null | tail -1 | show "w1m prune=$v pad=$pad"
done; done
unset OPENMM_HIP_PRUNE OPENMM_HIP_NL_PADDING
for v in 1 0; do
  export OPENMM_HIP_PRUNE=$v
  python bench.py --workload apoa1 --steps 1000 --warmup 100 --cpu-steps 0 --no-scale-workload --no-extra-workloads 2>/dev/null | tail -1 | show "apoa1 prune=$v"
done
unset OPENMM_HIP_PRUNE
for rep in 1 2 3; do for o in "" "--no-roofline"; do python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-steps 0 --no-scale-workload --no-extra-workloads $o 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('driver-line [$o]', d['value'], d['ms_per_step'], d.get('roofline',{}).get('avg_kernel_us'))"; done; done
