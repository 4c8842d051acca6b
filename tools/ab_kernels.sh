#!/bin/bash
# A/B of an environment switch inside ONE gpurun call, with the in-loop durations of chosen kernels:
#   bash tools/ab_kernels.sh VAR "A B C" REPS "kernel_a kernel_b" [bench flags]
VAR=$1; VALS=$2; REPS=$3; KERNELS=$4; shift 4
for k in $(seq 1 $REPS); do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 300 python bench.py --no-build --no-cpu-baseline --no-extra-lines --with-roofline --full-line --steps 24 --warmup 4 "$@" 2>/dev/null | KERNELS="$KERNELS" python -c "
import json,sys,os
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); ak=d['roofline']['all_kernels']
t=d['stage_thread_ms_per_step_concurrent']
print('$VAR=$v', d['value'], d['ms_per_step'], {k:round(x,1) for k,x in t.items()}, {k:ak[k]['avg_launch_us'] for k in ak if any(k.startswith(p) for p in os.environ['KERNELS'].split())})" || exit 1
  done
done
