#!/bin/bash
# conv1x1_resw timing ablations (lib: python att-aspp-unet_amd/build.py -DAAU_C3S_ABLATE --tag=abl): AAU_PW_ABL=k runs the normal launch
# AND the ablated one back to back, so (time with switch) - (time without) = the ablated kernel's own time
export AAU_LIB=$PWD/att-aspp-unet_amd/lib/libaau_abl.so
for abl in 0 2 4 6; do
  echo "== AAU_PW_ABL=$abl"
  AAU_PW_ABL=$abl timeout -k 10 200 python scripts/bench_kernels.py --only u1.up,u2.up,u2.gate --modes fwd,dgrad 2>&1 | grep -v "^totals\|grouped\|wgradL\|amdgpu.ids"
done
