#!/bin/bash
# HBM traffic per hand-written kernel of a configs[3] (DGCNN) episode: two rocprofv3 PMC passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_c4_$C
  rocprofv3 --pmc $C --kernel-include-regex "edgeconv|knn" --output-format csv -d $R/gpurun_out/pmc_c4_$C -o c4 -- \
    python3 $R/bench.py --workload c4 --no-extra --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_c4_$C.log 2>&1 || exit 1
done
python3 $R/tools/pmc_families.py $R/gpurun_out/pmc_c4_FETCH_SIZE $R/gpurun_out/pmc_c4_WRITE_SIZE 3
