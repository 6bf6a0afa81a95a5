#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv" 2>&1 | tail -3
for r in 1 2; do
for v in 0 1; do
  echo "== round $r NOLOADER=$v"
  if [ $v = 1 ]; then export AAU_C3_NOLOADER=1; else unset AAU_C3_NOLOADER; fi
  python scripts/bench_kernels.py --modes fwd,dgrad --only "d2.1" 2>&1 | grep -v amdgpu | head -1
  python scripts/bench_kernels.py --modes fwd,dgrad --only "d3.1" 2>&1 | grep -v amdgpu | head -1
  python scripts/bench_kernels.py --modes fwd,dgrad --only "d4.1" 2>&1 | grep -v amdgpu | head -1
  python scripts/bench_kernels.py --modes fwd,dgrad --only "u4.c0" 2>&1 | grep -v amdgpu | head -1
  python scripts/bench_kernels.py --modes fwd,dgrad --only "u3.c0" 2>&1 | grep -v amdgpu | head -1
  python scripts/bench_kernels.py --modes fwd,dgrad --only "u2.c0" 2>&1 | grep -v amdgpu | head -1
done; done
