#!/bin/bash
# round 4, visit 10d: the final tree -- HBM traffic of the pair / FFT launches (PMC passes), kernel statistics of the default bench and of
# amoeba2009 DHFR and AMOEBA water, the driver's bench line with the refreshed traffic
cd /root/repo; mkdir -p gpurun_out/r10d
TAG=r10d bash tools/gpu_visit.sh hbm trace:dhfr 2>&1 | tail -40
cd /tmp && export TMPDIR=/tmp
for w in "--dhfr" ""; do
  n=amoeba_water; [ -n "$w" ] && n=amoeba_dhfr
  timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r10d/prof -o $n -- python /root/repo/tools/bench_amoeba.py $w --steps 30 > /root/repo/gpurun_out/r10d/bench_$n.txt 2>&1
  ( cd /root/repo; python tools/rocpd_kernel_stats.py gpurun_out/r10d/prof/${n}_results.db > gpurun_out/r10d/${n}_kernel_stats.txt 2>&1; head -12 gpurun_out/r10d/${n}_kernel_stats.txt | cut -c1-150; tail -3 gpurun_out/r10d/${n}_kernel_stats.txt | cut -c1-200 )
  rm -rf /root/repo/gpurun_out/r10d/prof
done
cd /root/repo
timeout 600 python bench.py 2> gpurun_out/r10d/bench.err | tail -1 > gpurun_out/r10d/bench_driver_protocol.json
python -c "
import json; d = json.load(open('gpurun_out/r10d/bench_driver_protocol.json')); r = d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], r['traffic_source'][:120])"
cp profiles/pmc_pairs_fft.json gpurun_out/r10d/pmc_pairs_fft.json
