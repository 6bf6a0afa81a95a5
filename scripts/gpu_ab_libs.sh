#!/bin/bash
# same-box A/B of the whole step across library builds: gpu_ab_libs.sh <lib1.so> <lib2.so> ...   ("default" = libaau.so)
for i in 1 2 3; do
  for L in "$@"; do
    if [ "$L" = default ]; then E=""; else E="AAU_LIB=$L"; fi
    env $E python bench.py --no-cpu-baseline --no-infer --no-roofline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['ms_per_step'],3))" || exit 1
  done
done
