#!/bin/bash
# parity + A/B of the row-reuse 3x3 weight gradient (wgrad3x3r) against wgrad3x3
O=gpurun_out/w3r; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "wgrad" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
AAU_W3_NOR=1 timeout -k 10 300 python scripts/bench_kernels.py --modes wgrad > $O/old.txt 2>&1 && \
timeout -k 10 300 python scripts/bench_kernels.py --modes wgrad > $O/new.txt 2>&1
paste -d'\n' $O/old.txt $O/new.txt | grep -E "^(d1|d2|d3|d4|u4.c0|u3.c0|u2.c0|u1.c0|totals)"
