#!/bin/bash
# A/B an environment toggle inside ONE gpurun call: ab_env.sh "<modes>" "<layers>" VAR
modes=$1; layers=$2; var=$3
for round in 1 2; do
  for v in 0 1; do
    for l in $layers; do
      if [ $v = 1 ]; then export $var=1; else unset $var; fi
      echo -n "[$round] $var=$v "; python scripts/bench_kernels.py --modes $modes --only $l 2>&1 | grep -v amdgpu.ids | head -1
    done
  done
done
