#!/bin/bash
# A/B of several environment settings inside ONE gpurun call: bash tools/ab_multi.sh REPS "kernel prefixes" "A=1 B=2" "A=0" ...
# prints value / ms per step / stage threads / the in-loop average launch durations of the chosen kernels for every setting, REPS times in turn
REPS=$1; KERNELS=$2; shift 2
for k in $(seq 1 $REPS); do
  for setting in "$@"; do
    env $setting timeout -k 10 300 python bench.py --no-build --no-cpu-baseline --no-extra-lines --with-roofline --full-line --steps 24 --warmup 4 2>/dev/null | KERNELS="$KERNELS" SETTING="$setting" python -c "
import json,sys,os
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); ak=d['roofline']['all_kernels']
t=d['stage_thread_ms_per_step_concurrent']
print(os.environ['SETTING'], '|', d['value'], d['ms_per_step'], {k:round(x,1) for k,x in t.items()}, {k:ak[k]['avg_launch_us'] for k in ak if any(k.startswith(p) for p in os.environ['KERNELS'].split())})" || exit 1
  done
done
