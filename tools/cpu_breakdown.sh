#!/bin/bash
# per-thread-name CPU seconds per wall second of the timed loop, for one or more environment settings: bash tools/cpu_breakdown.sh "A=1" "A=0" [-- bench flags]
SETS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for s in "${SETS[@]}"; do
  ts=$(echo "$s" | tr ' ' '\n' | sed -n 's/^TASKSET=//p')
  env $s ${ts:+taskset -c $ts} timeout -k 10 300 python bench.py --no-build --no-cpu-baseline --no-extra-lines --steps 24 --warmup 4 --detail-out /tmp/cpu_detail.json "$@" > /dev/null 2>&1 || exit 1
  S="$s" python -c "
import json,os
d=json.load(open('/tmp/cpu_detail.json'))
h=d['config']['host_threads_gpu_path']
print('%-50s' % os.environ['S'], d['value'], 'cpu_s/s', h['cpu_s_per_wall_s_timed_region'], 'threads', h['threads_of_this_rank'])
print('   ', {k: v for k, v in h['cpu_s_per_wall_s_by_thread_name'].items()})"
done
