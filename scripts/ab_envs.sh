#!/bin/bash
# A/B env settings on bench_kernels.py inside ONE gpurun call: ab_envs.sh "<modes>" "VAR=a VAR2=b" "VAR=c" ...
modes=$1; shift
i=0
for round in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e python scripts/bench_kernels.py --modes $modes > gpurun_out/abe_${i}_$round.log 2>&1
    echo "[$round] {$e} $(tail -1 gpurun_out/abe_${i}_$round.log)"
  done
done
