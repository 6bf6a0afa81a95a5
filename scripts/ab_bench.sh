#!/bin/bash
# A/B full-step bench of several library builds inside ONE gpurun call (same device), alternating:
#   ab_bench.sh libA.so libB.so ...   -> ms_per_step and per-family ms for each
for round in 1 2; do
  for lib in "$@"; do
    AAU_LIB=$PWD/att-aspp-unet_amd/lib/$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$round] $lib', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['roofline']['ms_per_step_by_family'].items()})"
  done
done
