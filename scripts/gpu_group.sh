#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "group" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/bench_group.py 2>&1 | grep -v amdgpu.ids
