#!/bin/bash
# A/B of the XCD-aware workgroup order of the 3x3 weight gradients (AAU_W3_NOREMAP=1: round-1 order)
O=gpurun_out/w3map; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "wgrad" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
AAU_W3_NOREMAP=1 timeout -k 10 300 python scripts/bench_kernels.py --modes wgrad > $O/old.txt 2>&1 && \
timeout -k 10 300 python scripts/bench_kernels.py --modes wgrad > $O/new.txt 2>&1
paste -d'\n' $O/old.txt $O/new.txt | grep -E "^(d1|d2|d3|d4|u4.c0|u3.c0|u2.c0|u1.c0|totals)"
