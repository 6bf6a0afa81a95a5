#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels2_gpu.py tests/test_model_gpu.py tests/test_round2_api_gpu.py -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
bash scripts/gpu_ab_step.sh AAU_POOL_STORE_ROUTED=1
