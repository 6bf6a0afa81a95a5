#!/bin/bash
# parity + A/B of the persistent column-step kernel (conv3x3p; AAU_C3_NOPERSIST=1 = conv3x3h) against conv3x3h
O=gpurun_out/c3p; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python scripts/bench_c3fixed.py 2>&1 | grep Cin && \
timeout -k 10 200 python scripts/bench_c3fixed.py 2>&1 | grep Cin && \
timeout -k 10 300 python scripts/bench_kernels.py --modes fwd,dgrad > $O/h.txt 2>&1 && \
timeout -k 10 300 python scripts/bench_kernels.py --modes fwd,dgrad > $O/p.txt 2>&1
paste -d'\n' $O/h.txt $O/p.txt | grep -E "^(d2.1|d3|d4|u4.c0|u3.c0|u2.c0|totals)"
