#!/bin/bash
# round 2, visit v: kernel trace of the 8 serialized ranks (one rank's kernels at a time on the GPU): what a rank's step is made of
cd "$(dirname "$0")/.."
R=$(pwd); mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r2v -o trace -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29618 $R/bench.py --gpus 8 --steps 60 --warmup 5 --transport gloo --serialize-ranks --no-scale-workload --no-roofline --attempt-timeout 500 > $R/gpurun_out/prof_r2v.log 2>&1; echo "exit $?"
cd $R
ls -la gpurun_out/prof_r2v | head -30
for f in $(find gpurun_out/prof_r2v -name "*results.db" | head -40); do
  n=$(python tools/rocpd_kernel_stats.py $f 2>/dev/null | grep -c "nb_direct\|NbArgs")
  if [ "$n" -gt 0 ]; then echo "== $f"; python tools/rocpd_kernel_stats.py $f 2>&1 | head -24 | cut -c30-150; break; fi
done
