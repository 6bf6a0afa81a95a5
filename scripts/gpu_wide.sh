#!/bin/bash
# A/B of the 128x192 three-stage igemm tile on the bridge / ConvT GEMMs; usage: gpu_wide.sh <tag>
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wide or igemm" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
AAU_IGEMM_WIDE=0 timeout -k 10 300 python scripts/bench_kernels.py --only br. --modes fwd,dgrad > $O/narrow.txt 2>&1 && \
timeout -k 10 300 python scripts/bench_kernels.py --only br. --modes fwd,dgrad > $O/rule.txt 2>&1 && \
AAU_IGEMM_WIDE=1 timeout -k 10 300 python scripts/bench_kernels.py --only br. --modes fwd,dgrad > $O/wide.txt 2>&1
echo "--- narrow"; cat $O/narrow.txt; echo "--- rule"; cat $O/rule.txt; echo "--- wide"; cat $O/wide.txt
