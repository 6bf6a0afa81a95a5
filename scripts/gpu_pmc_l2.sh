#!/bin/bash
# L2 hit-rate pass over the real train step; usage: gpu_pmc_l2.sh <tag>
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-l2}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline --no-infer"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmcL -o runc -- $BENCH > $O/pmcL.log 2>&1; echo "pmcL rc=$?"
cd $R
python scripts/pmc_l2.py $O/pmcL $O/pmc_l2.txt | head -40
find $O/pmcL -name "*.csv" -size +20M -delete 2>/dev/null
