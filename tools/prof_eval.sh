#!/bin/bash
# rocprofv3 kernel trace of the evaluation loop (run on the GPU box) -> per-item steady-state kernel table.
#   bash tools/prof_eval.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
rm -rf $R/gpurun_out/prof_eval
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eval -o ev -- \
  python3 $R/tools/eval_loop.py 24 > $R/gpurun_out/prof_eval_$TAG.json 2> $R/gpurun_out/prof_eval_$TAG.err || exit 1
python3 $R/tools/steady_profile.py $R/gpurun_out/prof_eval 16 > $R/gpurun_out/steady_eval_$TAG.txt
head -30 $R/gpurun_out/steady_eval_$TAG.txt | cut -c1-150
