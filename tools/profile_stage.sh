#!/bin/bash
# Kernel trace + stats of one stage set of the loop: bash tools/profile_stage.sh TAG STAGES [bench flags] -> gpurun_out/prof_TAG/kernel_stats.csv
set -e
TAG=$1; STAGES=$2; shift 2
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --no-build --no-cpu-baseline --no-extra-lines --steps 12 --warmup 3 --stages $STAGES "$@" > $OUT/log 2>&1
cp $OUT/trace/bench_kernel_stats.csv $OUT/kernel_stats.csv
grep "^{" $OUT/log | cut -c1-400 > $OUT/line.txt || true
rm -rf $OUT/trace
