#!/bin/bash
# Quick GPU call: full GPU suite + the bench line + the per-launch table; usage: gpu_quick.sh <tag> [pytest-args]
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-q}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q ${2:-} > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --dump-launches $O/launches.txt > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 400 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value",d["value"],"ms",d["ms_per_step"],"frac",r["frac"],"traffic",r["traffic"],r.get("traffic_source"))
print(r["ms_per_step_by_family"])
print(d["cpu_baseline"])
PY
