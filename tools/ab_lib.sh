#!/bin/bash
# A/B of two BUILDS of the library inside ONE gpurun call (box-to-box spread is larger than most effects):
#   git stash; (cd tc2li-slam_amd/csrc && make); cp tc2li-slam_amd/lib/libtc2li_hip.so tools/scratch/libtc2li_base.so; git stash pop; (cd tc2li-slam_amd/csrc && make)
#   gpurun -- 'bash tools/ab_lib.sh bash tools/ab_one.sh'
# runs the command with the new, the base, the new and the base library in turn (tools/scratch/ is git-ignored but travels to the box).
cp tc2li-slam_amd/lib/libtc2li_hip.so /tmp/lib_new.so
for v in new base new base; do
  if [ $v = new ]; then cp /tmp/lib_new.so tc2li-slam_amd/lib/libtc2li_hip.so; else cp tools/scratch/libtc2li_base.so tc2li-slam_amd/lib/libtc2li_hip.so; fi
  echo "== $v"; "$@" || exit 1
done
cp /tmp/lib_new.so tc2li-slam_amd/lib/libtc2li_hip.so
