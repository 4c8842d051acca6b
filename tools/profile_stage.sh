#!/bin/bash
# Kernel statistics of one stage of the bench loop alone (diagnostic): bash tools/profile_stage.sh ba [rows] [sequences]   (through gpurun)
STAGE=${1:-ba}
EXTRA="${@:4}"  # further bench flags (e.g. --inertial-loop)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_stage
python __graft_entry__.py  # build un-profiled: no child process may start under rocprofv3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stage -o st -- python3 bench.py --no-build --no-extra-lines --no-cpu-baseline --steps 10 --warmup 2 --sequences ${3:-512} --stages $STAGE $EXTRA > gpurun_out/prof_stage.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_stage/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot/1e6,2))
for r in rows[:${2:-16}]: print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
rm -rf gpurun_out/prof_stage
