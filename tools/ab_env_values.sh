#!/bin/bash
# Same-box A/B of an environment variable between two VALUES on the default bench step (run on the GPU box):
#   bash tools/ab_env_values.sh FPSG_WINO_ROW_ALIGN 1 32 [rounds] [extra bench.py arguments ...]
V=$1; A=$2; B=$3; R=${4:-3}; shift 4
for i in $(seq $R); do
  for val in $A $B; do
    export $V=$val
    python3 bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 3 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$V=$val', round(d['value'],3), 'episodes/s', round(d['ms_per_step'],2), 'ms/step', d['clock'])"
  done
done
