#!/bin/bash
# Same-box A/B of two source trees: the working tree against a copy of an older one under build_exp/old
# (git archive <rev> fpsg_amd bench.py oracle | tar -x -C build_exp/old; cp fpsg_amd/libfpsg_hip.so build_exp/old/fpsg_amd/).
# usage (on the GPU box): tools/ab_tree.sh [workload] [rounds] [steps]
wl=${1:-c5}; rounds=${2:-3}; steps=${3:-10}
val() { python -c 'import json,sys; print(json.loads(sys.stdin.read())["value"])'; }
for i in $(seq $rounds); do
  a=$(python build_exp/old/bench.py --workload $wl --no-extra --no-cpu-baseline --steps $steps --warmup 3 2>/dev/null | val)
  b=$(python bench.py --workload $wl --no-extra --no-cpu-baseline --steps $steps --warmup 3 2>/dev/null | val)
  echo "$wl round $i: old $a  new $b"
done
