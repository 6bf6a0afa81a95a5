#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_fp16_gpu.py -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
echo "--- wide stores"; timeout -k 10 300 python scripts/bench_kernels.py --only u1.up,u2.up,u2.gate,u3.gate,u2.Wg --modes fwd,dgrad 2>&1 | grep "^u"
echo "--- 8-byte stores"; AAU_NO_WIDE_STORE=1 timeout -k 10 300 python scripts/bench_kernels.py --only u1.up,u2.up,u2.gate,u3.gate,u2.Wg --modes fwd,dgrad 2>&1 | grep "^u"
done
