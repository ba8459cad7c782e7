#!/bin/bash
# rocprofv3 kernel trace of the default bench step (run on the GPU box) -> per-episode steady-state kernel table.
#   bash tools/prof_step.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rm -rf $R/gpurun_out/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_step -o step -- \
  python3 $R/bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 2 "$@" > $R/gpurun_out/prof_step_$TAG.json 2> $R/gpurun_out/prof_step_$TAG.err || exit 1
python3 $R/tools/steady_profile.py $R/gpurun_out/prof_step 16 > $R/gpurun_out/steady_$TAG.txt
find $R/gpurun_out/prof_step -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kernel_stats_$TAG.csv \;
head -45 $R/gpurun_out/steady_$TAG.txt | cut -c1-150
