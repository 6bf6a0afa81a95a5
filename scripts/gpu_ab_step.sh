#!/bin/bash
# same-box A/B of the whole step: gpu_ab_step.sh "<ENV=1 ...>"   (variant B = the environment switches given)
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-infer --no-roofline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A default   ', round(d['ms_per_step'],3))" || exit 1
  env $1 python bench.py --no-cpu-baseline --no-infer --no-roofline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B $1', round(d['ms_per_step'],3))" || exit 1
done
