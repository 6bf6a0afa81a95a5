#!/bin/bash
# strip-kernel iteration: parity tests, then the micro-benchmark of the layers it serves (new kernel vs AAU_NO_C3S=1)
set -o pipefail
O=gpurun_out/${1:-c3s}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "strip or resident or two_plane" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/bench_kernels.py --only d1.1,d2.,u1.c0 --modes fwd,dgrad 2>&1 | grep -v "^totals\|grouped\|wgradL" | tee $O/bk_new.txt
if [ -z "$2" ]; then AAU_NO_C3S=1 timeout -k 10 300 python scripts/bench_kernels.py --only d1.1,d2.,u1.c0 --modes fwd,dgrad 2>&1 | grep -v "^totals\|grouped\|wgradL" | tee $O/bk_old.txt; fi
