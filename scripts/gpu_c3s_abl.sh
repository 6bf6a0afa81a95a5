#!/bin/bash
# strip-kernel timing ablations (lib built with: python att-aspp-unet_amd/build.py -DAAU_C3S_ABLATE --tag=abl)
# bits: 1 no MFMA (and no LDS reads), 2 fills out of range, 4 no stores, 8 no epilogue, 16 no barrier / wait, 32 no fill instructions, 64 MFMA without LDS reads
O=gpurun_out/${1:-c3sabl}
mkdir -p $O
export AAU_LIB=$PWD/att-aspp-unet_amd/lib/libaau_abl.so
for abl in ${ABLS:-0 1 6 8 14 16 32 46 47 63 64 70 78 110 126}; do
  echo "== AAU_C3S_ABL=$abl" | tee -a $O/abl.txt
  AAU_C3S_ABL=$abl timeout -k 10 200 python scripts/bench_kernels.py --only d1.1,d2.,u1.c0 --modes fwd 2>&1 | grep -v "^totals\|grouped\|wgradL\|amdgpu.ids" | tee -a $O/abl.txt
done
