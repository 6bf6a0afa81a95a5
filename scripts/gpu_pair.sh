#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_fp16_gpu.py -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
echo "--- paired"; timeout -k 10 300 python scripts/bench_kernels.py --only d1.1,d2.0,u1.c0 --modes fwd,dgrad 2>&1 | grep "^d\|^u"
echo "--- unpaired"; AAU_RESW_NOPAIR=1 timeout -k 10 300 python scripts/bench_kernels.py --only d1.1,d2.0,u1.c0 --modes fwd,dgrad 2>&1 | grep "^d\|^u"
