#!/bin/bash
# One measurement of tools/ab_lib.sh / tools/ab.sh: the Schur kernel of a 43-window lock-step group alone, then the whole loop with the
# instrumented pass (value, ms per step, the stage threads' times, a few kernels' average launch durations in the loop).
timeout -k 10 120 python tools/time_ba_kernels.py 43 2>&1 | grep -E "windows|schur_blocks"
timeout -k 10 300 python bench.py --no-build --no-cpu-baseline --no-extra-lines --with-roofline --full-line --steps 24 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); ak=d['roofline']['all_kernels']
print('value', d['value'], 'ms', d['ms_per_step'], d['stage_thread_ms_per_step_concurrent'])
print('   in loop us:', {k: round(ak[k]['avg_launch_us']) for k in ('k_ba_schur_blocks_b','k_ba_linearize_b','k_balm_hessian_b','k_fast_cells','k_blur7_strips','k_knn_plane') if k in ak}, 'sum', d['roofline']['kernel_ms_per_step_all_streams'])"
