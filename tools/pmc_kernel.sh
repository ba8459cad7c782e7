#!/bin/bash
# SQ counters of one kernel inside the default bench step (run on the GPU box):  bash tools/pmc_kernel.sh <kernel regex>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1
rm -rf $R/gpurun_out/pmc_kernel
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES \
  --kernel-include-regex "$K" --output-format csv -d $R/gpurun_out/pmc_kernel -o k -- \
  python3 $R/bench.py --no-extra --no-cpu-baseline --steps 1 --warmup 1 > $R/gpurun_out/pmc_kernel.log 2>&1 || { tail -5 $R/gpurun_out/pmc_kernel.log; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_kernel/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:50], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k[0]:50s} grid {k[1]:>9s} {k[2]:24s} {sum(v) / len(v):14.0f}  (n={len(v)})")
PY
