#!/bin/bash
# same-box A/B of the resw2 inner loop: column step (default build) vs one tap at a time (lib/libaau_taploop.so)
for i in 1 2 3; do
echo "--- column step"; timeout -k 10 200 python scripts/bench_kernels.py --only d1.1 --modes fwd,dgrad 2>&1 | grep "^d" || exit 1
echo "--- tap loop"; AAU_LIB=$PWD/att-aspp-unet_amd/lib/libaau_taploop.so timeout -k 10 200 python scripts/bench_kernels.py --only d1.1 --modes fwd,dgrad 2>&1 | grep "^d" || exit 1
done
