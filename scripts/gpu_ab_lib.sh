#!/bin/bash
# same-box A/B of the whole step between two library builds: gpu_ab_lib.sh <tag>   (B = lib/libaau_<tag>.so via AAU_LIB)
L=$PWD/att-aspp-unet_amd/lib/libaau_$1.so
[ -f $L ] || { echo "no $L"; exit 1; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-infer --no-roofline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A default   ', round(d['ms_per_step'],3))" || exit 1
  AAU_LIB=$L python bench.py --no-cpu-baseline --no-infer --no-roofline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B $1', round(d['ms_per_step'],3))" || exit 1
done
