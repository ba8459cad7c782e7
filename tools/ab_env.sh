#!/bin/bash
# Same-box A/B of an environment switch on the default bench step (run on the GPU box):
#   bash tools/ab_env.sh FPSG_STEM_FOLD [rounds]   -> episodes/s with the switch unset and =0, alternating
V=$1; R=${2:-3}
for i in $(seq $R); do
  for val in on off; do
    if [ $val = off ]; then export $V=0; else unset $V; fi
    python3 bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$V $val', round(d['value'],3), 'episodes/s', round(d['ms_per_step'],2), 'ms/step')"
  done
done
