#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
for abl in 0 2 6 10 14 4 8; do
  echo "--- ABL=$abl (2: no MFMA, 4: no activation fetch, 8: no weight fetch)"
  AAU_IGEMM_ABL=$abl timeout -k 10 200 python scripts/bench_kernels.py --only br.d6,br.proj --modes fwd 2>&1 | grep "^br" || exit 1
done
