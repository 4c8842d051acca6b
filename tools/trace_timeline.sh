#!/bin/bash
# Kernel trace of a short bench run, kept whole (gzip) for timeline analysis with tools/timeline.py.
# Usage: bash tools/trace_timeline.sh <tag> [bench flags]
set -e
TAG=${1:-tl}
shift || true
OUT=gpurun_out/timeline_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python __graft_entry__.py
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python3 bench.py --no-build --no-cpu-baseline --no-extra-lines --steps 6 --warmup 2 $* > $OUT/bench.log 2>&1
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/timeline.py $f > $OUT/timeline.txt
gzip -c $f > $OUT/kernel_trace.csv.gz
rm -rf $OUT/trace
tail -40 $OUT/timeline.txt
