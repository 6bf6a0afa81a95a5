#!/bin/bash
# A/B whole-table runs of bench_kernels.py for several library builds inside ONE gpurun call:
#   ab_libs.sh "<modes>" libA libB ...   (two alternating rounds; logs under gpurun_out/ab_<lib>_<round>.log)
modes=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    AAU_LIB=$PWD/att-aspp-unet_amd/lib/$lib python scripts/bench_kernels.py --modes $modes > gpurun_out/ab_${lib%.so}_$round.log 2>&1
    echo "[$round] $lib $(tail -1 gpurun_out/ab_${lib%.so}_$round.log)"
  done
done
