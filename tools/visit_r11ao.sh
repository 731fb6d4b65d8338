#!/bin/bash
# round 5, visit ao: timeline of one AMOEBA DHFR step (kernel trace with start times per queue)
cd "$(dirname "$0")/.."
R=$(pwd)
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o trace -- python $R/tools/bench_amoeba.py --dhfr --steps 12 > $R/gpurun_out/r11ao_traced.log 2>&1 )
f=$(find gpurun_out/prof_tl -name "*.db" | head -1)
[ -n "$f" ] && python tools/rocpd_timeline.py "$f" 420 k_mp_frames > gpurun_out/r11ao_amoeba_dhfr_timeline.txt 2>&1
rm -rf gpurun_out/prof_tl
wc -l gpurun_out/r11ao_amoeba_dhfr_timeline.txt
