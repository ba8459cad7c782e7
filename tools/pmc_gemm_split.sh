#!/bin/bash
# SQ counters of K10 (run on the GPU box): where a wave's cycles go.  bash tools/pmc_gemm_split.sh "<bench args>" [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS=${1:---quick --variants 2 --packed 0 --rounds 1 --reps 3}
TAG=${2:-gs}
for pass in 1 2 3; do
  case $pass in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES";;
    2) C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE";;
    3) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT";;
  esac
  rm -rf $R/gpurun_out/pmc_${TAG}_$pass
  rocprofv3 --pmc $C --kernel-include-regex "gemm_split_(pa_|pnn_)?kernel" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$pass -o k -- \
    python3 $R/tools/bench_gemm_split.py $ARGS > $R/gpurun_out/pmc_${TAG}_$pass.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_$pass.log; }
done
python3 - <<PY
import csv, glob, collections, re
for p in (1, 2, 3):
    acc = collections.defaultdict(list)
    for f in glob.glob("$R/gpurun_out/pmc_${TAG}_%d/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"fpsg::\(anonymous namespace\)::", "", r["Kernel_Name"])[:64]
            acc[(name, r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        v = acc[k]
        print(f"{k[0]:64s} grid {k[1]:>9s} {k[2]:26s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
