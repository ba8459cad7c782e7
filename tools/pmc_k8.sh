#!/bin/bash
# HBM-side counters of K8 (tools/bench_k8.py) on the GPU box: FETCH_SIZE, WRITE_SIZE, L2 hits / misses, in separate passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  tag=$(echo $C | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_k8_$tag
  rocprofv3 --pmc $C --kernel-include-regex conv_first_dw_kernel --output-format csv -d $R/gpurun_out/pmc_k8_$tag -o k -- \
    python3 $R/tools/bench_k8.py > $R/gpurun_out/pmc_k8_$tag.log 2>&1 || { tail -3 $R/gpurun_out/pmc_k8_$tag.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_k8_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k[0]:70s} {k[1]:22s} {sum(v) / len(v):16.1f}  (n={len(v)})")
PY
