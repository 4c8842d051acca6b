#!/bin/bash
# the MFMA line alone: 3 lock-step groups (as benched) and 1 group (the product kernel alone on the GPU: 96 windows per launch); $1: env settings to compare
for setting in "${@:-X=0}"; do
for g in 3 1; do
env $setting TC2LI_BA_LOCKSTEP_GROUPS=$g python bench.py --no-build --no-cpu-baseline --mfma-only 2>/dev/null | tail -1 | G=$g S="$setting" python -c "
import json,sys,os; d=json.loads(sys.stdin.read())['mfma_config']; r=d['roofline']
print(os.environ['S'], 'groups', os.environ['G'], 'windows/s', d['windows_per_s'], 'ms/batch', d['ms_per_batch'], 'launch ms', r['avg_launch_ms'], 'launches', r['launches'], 'TFLOP/s', r['achieved'], 'frac', r['frac'], 'useful', r['useful_frac'])"
done
done
