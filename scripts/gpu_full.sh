#!/bin/bash
# full GPU suite + bench line (no CPU baseline) ; usage: gpu_full.sh <tag>
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-infer --dump-launches $O/launches.txt > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value",round(d["value"],1),"ms",round(d["ms_per_step"],3),"fam",{k:round(v,3) for k,v in r["ms_per_step_by_family"].items()})
for k,v in list(r["by_kernel"].items())[:12]: print(f"  {k:28s} n={v['launches_per_step']:5.1f} ms={v['ms_per_step']:.3f} TF={v['tflops']:7.1f}")
PY
