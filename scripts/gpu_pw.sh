#!/bin/bash
# resident-weight 1x1 kernel (ConvT forward, gate convs): parity tests, then micro-benchmark ring of 7 tiles vs the round-2 double buffer
set -o pipefail
O=gpurun_out/${1:-pw}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_kernels2_gpu.py tests/test_blocks_gpu.py -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for env in "X=1" "AAU_PW_NOSTAGE=1" "AAU_PW_NBUF=2"; do
  echo "== $env"
  env $env timeout -k 10 300 python scripts/bench_kernels.py --only u1.up,u2.up,u2.gate,u3.gate,u2.Wg --modes fwd,dgrad 2>&1 | grep -v "^totals\|grouped\|wgradL\|amdgpu.ids" | tee $O/bk_$env.txt
done
