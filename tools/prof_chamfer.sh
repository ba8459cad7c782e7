#!/bin/bash
# rocprofv3 kernel trace of the K1 micro-benchmark (run on the GPU box): per-kernel durations.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_k1
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k1 -- python3 $GRAFT_REPO_ROOT/tools/bench_chamfer.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_k1.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_k1_kernel_stats.csv \;
cat $GRAFT_REPO_ROOT/gpurun_out/prof_k1_kernel_stats.csv | cut -c1-200
