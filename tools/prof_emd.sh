#!/bin/bash
# rocprofv3 kernel stats of the K2 / K2b micro-benchmark (run on the GPU box): per-kernel durations of the sweeps.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_emd
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_emd -o emd -- \
  python3 $R/tools/bench_emd.py > $R/gpurun_out/prof_emd.txt 2> $R/gpurun_out/prof_emd.err || exit 1
find $R/gpurun_out/prof_emd -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kernel_stats_emd.csv \;
cat $R/gpurun_out/prof_emd.txt; cut -c1-220 $R/gpurun_out/kernel_stats_emd.csv | head -20
