#!/bin/bash
# timing-only ablations of the resident-weight 1x1 kernel (AAU_PW_ABL: 1 = no stores, 2 = no MFMAs)
O=gpurun_out/pw; mkdir -p $O
for abl in 0 1 2 3; do
  AAU_PW_ABL=$abl timeout -k 10 300 python scripts/bench_kernels.py --modes fwd,dgrad --only up > $O/abl$abl.txt 2>&1 && \
  AAU_PW_ABL=$abl timeout -k 10 300 python scripts/bench_kernels.py --modes fwd,dgrad --only gate >> $O/abl$abl.txt 2>&1
  echo "abl=$abl"; grep -E "u1.up|u2.up|u2.gate|u3.gate" $O/abl$abl.txt
done
