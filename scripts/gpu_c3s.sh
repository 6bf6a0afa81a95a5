#!/bin/bash
# strip-kernel iteration: parity tests (default, AAU_C3S_NOSTAG=1, AAU_C3S_MODE=2), then the micro-benchmark A/B
set -o pipefail
O=gpurun_out/${1:-c3s}
mkdir -p $O
for env in "X=1" "AAU_C3S_NOSTAG=1" "AAU_C3S_MODE=2"; do
  env $env timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "strip or resident or two_plane" > $O/pytest_$env.log 2>&1; rc=$?; tail -2 $O/pytest_$env.log
  [ $rc -ne 0 ] && exit $rc
done
for env in "X=1" "AAU_C3S_NOSTAG=1" "X=2" "AAU_C3S_NOSTAG=2"; do
  echo "== $env"
  env $env timeout -k 10 300 python scripts/bench_kernels.py --only d1.1,d2.,u1.c0 --modes fwd,dgrad 2>&1 | grep -v "^totals\|grouped\|wgradL\|amdgpu.ids" | tee $O/bk_$env.txt
done
