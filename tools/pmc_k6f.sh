#!/bin/bash
# SQ counters of K6f (run on the GPU box): where a wave's cycles go (issue / parked at s_waitcnt / issue stalls).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_k6f
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-include-regex wino4_fused --output-format csv -d $R/gpurun_out/pmc_k6f -o k -- \
  python3 $R/tools/bench_k6f.py > $R/gpurun_out/pmc_k6f.log 2>&1 || { tail -5 $R/gpurun_out/pmc_k6f.log; exit 1; }
rm -rf $R/gpurun_out/pmc_k6f2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  --kernel-include-regex wino4_fused --output-format csv -d $R/gpurun_out/pmc_k6f2 -o k -- \
  python3 $R/tools/bench_k6f.py > $R/gpurun_out/pmc_k6f2.log 2>&1 || { tail -5 $R/gpurun_out/pmc_k6f2.log; }
python3 - <<PY
import csv, glob, collections
for d in ("pmc_k6f", "pmc_k6f2"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        v = acc[k]
        print(f"grid {k[0]:>8s} {k[1]:28s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
