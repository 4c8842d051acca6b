#!/bin/bash
# A/B of two BUILDS of the library inside ONE gpurun call, with the in-loop durations of chosen kernels (tools/ab_lib.sh for how the base
# library gets into tools/scratch/): bash tools/ab_lib_kernels.sh REPS "kernel prefixes"
REPS=$1; KERNELS=$2
cp tc2li-slam_amd/lib/libtc2li_hip.so /tmp/lib_new.so
for k in $(seq 1 $REPS); do
  for v in new base; do
    if [ $v = new ]; then cp /tmp/lib_new.so tc2li-slam_amd/lib/libtc2li_hip.so; else cp tools/scratch/libtc2li_base.so tc2li-slam_amd/lib/libtc2li_hip.so; fi
    bash tools/ab_multi.sh 1 "$KERNELS" "TC2LI_AB_LIB=$v" || exit 1
  done
done
cp /tmp/lib_new.so tc2li-slam_amd/lib/libtc2li_hip.so
