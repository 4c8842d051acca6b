#!/bin/bash
# A/B of several environment SETTINGS inside one gpurun call: bash tools/ab_env.sh REPS "A=1 B=2" "A=0" ... [-- bench flags]
# prints value / ms_per_step and the stage threads' times of every run, the settings in turn, REPS times
REPS=$1; shift
SETS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for k in $(seq 1 $REPS); do
  for s in "${SETS[@]}"; do
    ts=$(echo "$s" | tr ' ' '\n' | sed -n 's/^TASKSET=//p')   # TASKSET=0-7 among the settings: the run confined to those CPUs (before any GPU call)
    env $s ${ts:+taskset -c $ts} timeout -k 10 300 python bench.py --no-build --no-cpu-baseline --no-extra-lines --steps 24 --warmup 4 "$@" 2>/dev/null | S="$s" python -c "
import json,sys,os
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
t=d.get('stage_thread_ms_per_step_concurrent') or {}
h=(d.get('config') or {}).get('host') or (d.get('config') or {}).get('host_threads_gpu_path') or {}
print('%-44s' % os.environ['S'], d.get('value'), d['ms_per_step'], 'cpu_s/s', h.get('cpu_s_per_wall_s_timed_region'), 'threads', h.get('threads_of_this_rank'), {k[:-7]: round(v,1) for k,v in t.items()})" || exit 1
  done
done
