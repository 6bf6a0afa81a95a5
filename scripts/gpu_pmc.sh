#!/bin/bash
# one PMC pass (MFMA / LDS / wait counters) + one kernel trace of the real step; usage: gpu_pmc.sh <tag>
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-pmc}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline --no-infer"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- $BENCH > $O/kt.log 2>&1; echo "kt rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmcM -o runc -- $BENCH > $O/pmcM.log 2>&1; echo "pmcM rc=$?"
cd $R
python scripts/analyze_trace.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/kernel_summary.txt 2>&1
python scripts/pmc_mfma.py $O/pmcM $O/pmc_mfma.txt > /dev/null 2>&1; echo "mfma rc=$?"
find $O/kt $O/pmcM -name "*.csv" -size +20M -delete 2>/dev/null
head -40 $O/pmc_mfma.txt
