#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "wgrad" 2>&1 | tail -3
python scripts/bench_kernels.py --modes wgrad 2>&1 | grep -v amdgpu
