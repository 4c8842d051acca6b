#!/bin/bash
# A/B of an environment switch inside ONE gpurun call (box-to-box spread is larger than most effects): bash tools/ab.sh VAR A B [reps] [bench flags]
# prints value / ms_per_step and the stage threads' times of every run, alternating A B A B ...
VAR=$1; A=$2; B=$3; REPS=${4:-2}; shift 4
for k in $(seq 1 $REPS); do
  for v in "$A" "$B"; do
    env $VAR=$v timeout -k 10 300 python bench.py --no-build --no-cpu-baseline --no-extra-lines --steps 24 --warmup 4 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$VAR=$v', d['value'], d['ms_per_step'], d['stage_thread_ms_per_step_concurrent'])" || exit 1
  done
done
