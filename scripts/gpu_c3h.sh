#!/bin/bash
# A/B of the column-step halo kernel (conv3x3h) against the grouped-tap kernel (conv3x3g)
O=gpurun_out/c3h; mkdir -p $O
AAU_C3_H=1 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_kernels2_gpu.py tests/test_blocks_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout -k 10 300 python scripts/bench_kernels.py --modes fwd,dgrad > $O/g.txt 2>&1 && \
AAU_C3_H=1 timeout -k 10 300 python scripts/bench_kernels.py --modes fwd,dgrad > $O/h.txt 2>&1
paste -d'\n' $O/g.txt $O/h.txt | grep -E "^(d2.1|d3|d4|u4.c0|u3.c0|u2.c0|totals)"
