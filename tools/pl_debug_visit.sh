#!/bin/bash
# profiling visit for the AMOEBA pair-list builder: SQ counters of pl_build (two passes), direct polarization (3 steps)
cd "$(dirname "$0")/.."
R=$(pwd)
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_pl -o pmc -- python $R/tools/bench_amoeba.py --steps 2 --direct > /dev/null 2>&1 )
  f=$(find gpurun_out/pmc_pl -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import sys, csv, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    for name in ("pl_build", "k_mp_forces", "k_vdw_pairs_list", "k_mp_field"):
        if name in k:
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[(name, r["Counter_Name"])] += 1
for name in acc:
    print(name, {c: round(v / n[(name, c)]) for c, v in acc[name].items()}, "calls", max(n[(name, c)] for c in acc[name]))
PY
  rm -rf gpurun_out/pmc_pl
done 2>&1 | tee gpurun_out/${TAG:-r5o}_pl_build_sq_counters.txt
