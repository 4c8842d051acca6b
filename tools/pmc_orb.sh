#!/bin/bash
# SQ counters of the ORB kernels alone (one chunk: the kernels one after the other): bash tools/pmc_orb.sh -> gpurun_out/pmc_orb/*.csv summaries
set -e
OUT=gpurun_out/pmc_orb; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TC2LI_ORB_CHUNKS=1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_IFETCH" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o t -- python3 tools/time_orb_alone.py 256 > $OUT/log$i 2>&1 || { tail -5 $OUT/log$i; continue; }
  python3 - "$OUT/p$i" "$set" <<'PY' >> $OUT/summary.txt
import csv, sys, glob, collections
d, names = sys.argv[1], sys.argv[2].split()
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name'].split('(')[0][:40]
    agg[k][row['Counter_Name']] += float(row['Counter_Value']); 
    cnt[(k, row['Counter_Name'])] += 1
for k in agg:
    if not any(s in k for s in ('blur', 'fast_cells', 'orient', 'resize', 'quadtree_sorted_list')): continue
    print(k, {n: round(agg[k][n] / max(cnt[(k, n)], 1), 1) for n in names})
PY
  rm -rf $OUT/p$i
done
cat $OUT/summary.txt
