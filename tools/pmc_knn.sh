#!/bin/bash
# SQ counters of the K3 kernels inside tools/bench_knn.py (run on the GPU box):  bash tools/pmc_knn.sh [kernel regex]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-knn_stream_kernel}
rm -rf $R/gpurun_out/pmc_knn
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS \
  --kernel-include-regex "$K" --output-format csv -d $R/gpurun_out/pmc_knn/a -o k -- \
  python3 $R/tools/bench_knn.py 2 > $R/gpurun_out/pmc_knn_a.log 2>&1 || { tail -5 $R/gpurun_out/pmc_knn_a.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM \
  --kernel-include-regex "$K" --output-format csv -d $R/gpurun_out/pmc_knn/b -o k -- \
  python3 $R/tools/bench_knn.py 2 > $R/gpurun_out/pmc_knn_b.log 2>&1 || { tail -5 $R/gpurun_out/pmc_knn_b.log; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_knn/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k[0]:60s} grid {k[1]:>9s} {k[2]:24s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
