#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (run on the GPU box): the per-kernel durations the line's
# roofline numbers must agree with.   bash tools/prof_default_bench.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
rm -rf $R/gpurun_out/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -o bench -- \
  python3 $R/bench.py > $R/gpurun_out/bench_default_under_rocprof_$TAG.json 2> $R/gpurun_out/bench_default_under_rocprof_$TAG.err || exit 1
find $R/gpurun_out/prof_default -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/bench_default_kernel_stats_$TAG.csv \;
grep -i "chamfer" $R/gpurun_out/bench_default_kernel_stats_$TAG.csv | cut -c1-60,150-400
rm -rf $R/gpurun_out/prof_default
