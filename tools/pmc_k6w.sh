#!/bin/bash
# SQ / TCP counters of K6w (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_k6w
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-include-regex wino4_dw --output-format csv -d $R/gpurun_out/pmc_k6w -o k -- \
  python3 $R/tools/bench_k6f.py > $R/gpurun_out/pmc_k6w.log 2>&1 || { tail -5 $R/gpurun_out/pmc_k6w.log; exit 1; }
rm -rf $R/gpurun_out/pmc_k6w2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum \
  --kernel-include-regex wino4_dw --output-format csv -d $R/gpurun_out/pmc_k6w2 -o k -- \
  python3 $R/tools/bench_k6f.py > $R/gpurun_out/pmc_k6w2.log 2>&1 || { tail -5 $R/gpurun_out/pmc_k6w2.log; }
python3 - <<PY
import csv, glob, collections
for d in ("pmc_k6w", "pmc_k6w2"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        v = acc[k]
        print(f"grid {k[0]:>8s} {k[1]:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
