#!/bin/bash
# Same-box A/B of two GEMM record files on the default bench step (run on the GPU box):
#   bash tools/ab_records.sh <records A> <records B> [rounds] [extra bench.py arguments ...]
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq $R); do
  for f in $A $B; do
    FPSG_GEMM_TUNING_FILE=$f python3 bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 3 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$f', round(d['value'],3), 'episodes/s', round(d['ms_per_step'],2), 'ms/step', d['clock']['sclk_mhz_under_last_warmup_step'])"
  done
done
