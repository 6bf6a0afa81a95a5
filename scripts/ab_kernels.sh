#!/bin/bash
# A/B several library builds inside ONE gpurun call (same device): usage ab_kernels.sh "<modes>" "<layers>" libA libB ...
modes=$1; layers=$2; shift 2
for round in 1 2; do
  for lib in "$@"; do
    for l in $layers; do
      echo -n "[$round] $(basename $lib) "; AAU_LIB=$PWD/att-aspp-unet_amd/lib/$lib python scripts/bench_kernels.py --modes $modes --only $l 2>&1 | grep -v amdgpu.ids | head -1
    done
  done
done
