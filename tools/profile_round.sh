#!/bin/bash
# Round profile on the GPU box: kernel trace + stats of the default bench command, then the HBM counters of the same
# command in two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; no trace domains next to --pmc).
# Usage (from the repo root, through gpurun): bash tools/profile_round.sh r02 [extra bench flags]
# The summaries land in gpurun_out/profiles_<tag>/ (gpurun only brings gpurun_out/ back): copy them into profiles/ afterwards.
set -e
TAG=${1:-r02}
shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# Build BEFORE any rocprofv3 line: under the profiler's preload a child process (make -> sh -> hipcc) would be an exec from a process
# that has initialised the GPU.  bench.py --no-build then fails instead of building, and the command after `--` stays one interpreter.
python __graft_entry__.py
ARGS="bench.py --no-build --no-cpu-baseline --steps 24 --warmup 4 $*"
# --no-extra-lines: the process then runs the default workload only (warm-up, timed steps, the instrumented pass), so the CSV's per-kernel
# averages are averages over the same loop the line's roofline is measured in
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ARGS --no-extra-lines --with-roofline > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $ARGS --no-extra-lines > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python3 $ARGS --no-extra-lines > $OUT/bench_write.log 2>&1
# matrix-unit occupancy of the dense Schur GEMM: cycles the MFMA pipe is busy next to the cycles its waves exist (own pass; SQ counters).
# Round 4: the pass profiles `bench.py --mfma-only` -- the 25-keyframe bLarge LocalLVIBA batch, the workload that runs the MFMA kernel
# (the default loop's Schur product is a vector-unit kernel since round 3 and issues no matrix instruction)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -o bench -- python3 bench.py --no-build --no-cpu-baseline --mfma-only > $OUT/bench_mfma.log 2>&1
# HBM counters of the inertial loop (configs[3]): inertial_config.roofline.traffic
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_i -o bench -- python3 $ARGS --no-extra-lines --inertial-loop > $OUT/bench_fetch_i.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_i -o bench -- python3 $ARGS --no-extra-lines --inertial-loop > $OUT/bench_write_i.log 2>&1
# kernel trace + stats of the inertial loop (configs[3] with the bLarge LocalLVIBA windows: k_lvi_solve_b, k_ba_schur_full_b, the time sort ...)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_i -o bench -- python3 $ARGS --no-extra-lines --inertial-loop > $OUT/bench_trace_i.log 2>&1
# instruction mix and LDS behaviour of every kernel (the bound of k_fast_cells is stated from these): own passes, SQ counters only
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_insts -o bench -- python3 $ARGS --no-extra-lines > $OUT/bench_insts.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_lds -o bench -- python3 $ARGS --no-extra-lines > $OUT/bench_lds.log 2>&1
# round 6: the small-batch regime and the extraction alone, kernel trace + stats each
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_64 -o bench -- python3 bench.py --no-build --no-cpu-baseline --no-extra-lines --sequences 64 --steps 40 --warmup 8 > $OUT/bench_trace_64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_orb -o bench -- python3 bench.py --no-build --no-cpu-baseline --no-extra-lines --front-end-only --stages orb --steps 12 --warmup 3 > $OUT/bench_trace_orb.log 2>&1
# every pass must have left through a normal exit: no abort hidden behind the profiler's signal handler
if grep -l "signal 6\|terminate called\|Memory access fault" $OUT/*.log; then echo "profile_round: a profiled process died (see the logs above)"; exit 1; fi
python tools/summarize_profile.py $OUT $TAG
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
# keep the merge small: the raw traces stay on the box
rm -rf $OUT/trace $OUT/trace_i $OUT/trace_64 $OUT/trace_orb $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/pmc_insts $OUT/pmc_lds $OUT/pmc_fetch_i $OUT/pmc_write_i
