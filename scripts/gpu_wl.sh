#!/bin/bash
for f in 0 3 4; do
  echo "== flags $f"; AAU_WL_FLAGS=$f python scripts/bench_wgradL.py 2>&1 | grep -v amdgpu.ids | head -3
done
python scripts/bench_kernels.py --only br. --modes wgrad 2>&1 | tail -1
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "wgrad_group" 2>&1 | tail -1
