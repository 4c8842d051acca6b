#!/bin/bash
# the configs[3] single-sequence leg as bench.py's child runs it: bash tools/single_inertial.sh [REPS] [extra env]
for k in $(seq 1 ${1:-2}); do
  GPU_MAX_HW_QUEUES=24 TC2LI_NO_BUILD=1 python bench.py --gpus 1 --sequences 1 --unique 1 --steps 200 --warmup 300 --no-cpu-baseline --no-extra-lines --no-build --inertial-loop 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs[3], one sequence:', d['value'], 'frames/s', d.get('stage_thread_ms_per_step_concurrent'))"
done
