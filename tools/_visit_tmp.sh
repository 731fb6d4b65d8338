timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -2
show() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], 'rows', r.get('rows'), r.get('rows_as_built'), 'rebuilds', r.get('rebuilds'), {k:(round(v['avg_us'],1) if v['avg_us'] else None) for k,v in r.get('kernel_timers_us',{}).items()}, r.get('fp32_issue',{}).get('evals_per_useful_pair'))"; }
for rep in 1 2; do for v in prune noprune; do
  if [ $v = noprune ]; then export OPENMM_HIP_NO_PRUNE=1; else unset OPENMM_HIP_NO_PRUNE; fi
  python bench.py --steps 3000 --warmup 300 --cpu-steps 0 --no-scale-workload --no-extra-workloads 2>/dev/null | tail -1 | show "dhfr $v"
done; done
for v in prune noprune; do
  if [ $v = noprune ]; then export OPENMM_HIP_NO_PRUNE=1; else unset OPENMM_HIP_NO_PRUNE; fi
  python bench.py --workload water1m --steps 300 --warmup 20 --cpu-steps 0 --no-scale-workload --no-extra-workloads 2>/dev/null | tail -1 | show "w1m $v"
done
unset OPENMM_HIP_NO_PRUNE
for rep in 1 2 3; do for o in "" "--no-roofline"; do python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-steps 0 --no-scale-workload --no-extra-workloads $o 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('driver-line [$o]', d['value'], d['ms_per_step'], d.get('roofline',{}).get('avg_kernel_us'))"; done; done
