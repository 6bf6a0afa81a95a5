#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_fp16_gpu.py tests/test_blocks_gpu.py -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
L=d1.1,d2.0,d2.1,d3.1,d4.1,u4.c0,u2.c0,u1.c0
for i in 1 2; do
echo "--- wide stores"; timeout -k 10 300 python scripts/bench_kernels.py --only $L --modes fwd,dgrad 2>&1 | grep "^[du][1-4]"
echo "--- 8-byte stores"; AAU_NO_WIDE_STORE=1 timeout -k 10 300 python scripts/bench_kernels.py --only $L --modes fwd,dgrad 2>&1 | grep "^[du][1-4]"
done
