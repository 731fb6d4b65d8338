cd /root/repo
TAG=r07g_hs bash tools/gpu_visit.sh serial:8 2>&1 | tail -1
OPENMM_HIP_DD_CUBE_CURVE=1 TAG=r07g_hs_cube bash tools/gpu_visit.sh serial:8 2>&1 | tail -1
OPENMM_HIP_DD_BOTH_SIDES=1 TAG=r07g_bs bash tools/gpu_visit.sh serial:8 2>&1 | tail -1
OPENMM_HIP_NL_TRACE=1 TAG=r07g_trace bash tools/gpu_visit.sh serial:8 2>&1 | tail -1
grep -h "nl_find trace" gpurun_out/r07g_trace_serialized_n8.err | head -6
TAG=r07g_hs bash tools/gpu_visit.sh serialtrace:8 2>&1 | tail -22 | cut -c1-200
bash tools/gpu_visit.sh tests:"tests/test_gpu_multirank.py" 2>&1 | tail -4
