cd /root/repo
export TAG=r07b
bash tools/gpu_visit.sh tests:"tests/test_gpu_platform.py -k moeba" 2>&1 | tail -6
for cfg in "-" "OPENMM_HIP_AMOEBA_PRECISION=double" "OPENMM_HIP_AMOEBA_VDW_MAIN_STREAM=1" "OPENMM_HIP_AMOEBA_PRECISION=double,OPENMM_HIP_AMOEBA_VDW_MAIN_STREAM=1" "OPENMM_HIP_AMOEBA_SKIN=0" "OPENMM_HIP_AMOEBA_SKIN=0.03" "OPENMM_HIP_AMOEBA_PREDICTOR_POINTS=6"; do
  ( if [ "$cfg" != "-" ]; then for kv in ${cfg//,/ }; do export "$kv"; done; fi
    echo "== $cfg"; timeout 200 python tools/bench_amoeba.py --steps 40 2>&1 | tail -1 | cut -c1-420 ) 2>&1 | tee -a gpurun_out/${TAG}_amoeba_ab.txt
done
timeout 200 python tools/bench_amoeba.py --steps 40 --direct 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_amoeba_ab.txt
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_amo -o trace -- python $R/tools/bench_amoeba.py --steps 20 > /dev/null 2>&1 )
if [ -f gpurun_out/prof_amo/trace_results.db ]; then python tools/rocpd_kernel_stats.py gpurun_out/prof_amo/trace_results.db > gpurun_out/${TAG}_amoeba_water_kernel_stats.txt 2>&1
else f=$(find gpurun_out/prof_amo -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${TAG}_amoeba_water_kernel_stats.txt; fi
head -34 gpurun_out/${TAG}_amoeba_water_kernel_stats.txt | cut -c1-160
rm -rf gpurun_out/prof_amo
