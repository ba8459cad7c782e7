#!/bin/bash
# HBM traffic of the K1 ops inside the default bench step (run on the GPU box): two rocprofv3 PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share one), summarised into gpurun_out/k1_traffic.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_k1_$C
  rocprofv3 --pmc $C --kernel-include-regex chamfer --output-format csv -d $R/gpurun_out/pmc_k1_$C -o k1 -- \
    python3 $R/bench.py --no-extra --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_k1_$C.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_k1_FETCH_SIZE $R/gpurun_out/pmc_k1_WRITE_SIZE $R/gpurun_out/k1_traffic.json | tail -30
