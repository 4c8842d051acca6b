#!/bin/bash
# the loop at a small batch under several environment settings, with the sweep child's flags: bash tools/ab_small.sh SEQUENCES REPS "A=1" "B=2" ...
N=$1; REPS=$2; shift 2
STEPS=$(( 5120 / N )); [ $STEPS -gt 40 ] && STEPS=40; [ $STEPS -lt 10 ] && STEPS=10
for k in $(seq 1 $REPS); do
  for s in "$@"; do
    env $s timeout -k 10 300 python bench.py --no-build --no-cpu-baseline --no-extra-lines --sequences $N --steps $STEPS --warmup 8 2>/dev/null | S="$s" python -c "
import json,sys,os
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
t=d.get('stage_thread_ms_per_step_concurrent') or {}
print('%-60s' % os.environ['S'], d.get('value'), d['ms_per_step'], {k[:-7]: round(v,2) for k,v in t.items()})" || exit 1
  done
done
