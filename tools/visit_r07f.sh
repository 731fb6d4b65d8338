cd /root/repo
OPENMM_HIP_NL_TRACE=1 OPENMM_HIP_DD_DRIFT=0.75 OPENMM_HIP_REORDER_LAG=128 TAG=r07f_trace bash tools/gpu_visit.sh serial:8 2>&1 | tail -1
grep -h "nl_find trace" gpurun_out/r07f_trace_serialized_n8.err | head -30
OPENMM_HIP_NL_TRACE=1 timeout 600 python bench.py --workload water1m --steps 40 --warmup 5 --cpu-steps 0 --no-roofline --no-extra-workloads 2> gpurun_out/r07f_single_trace.err | tail -1 | cut -c1-300
grep -h "nl_find trace" gpurun_out/r07f_single_trace.err | head -8
