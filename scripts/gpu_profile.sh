#!/bin/bash
# Profile call: full GPU suite, bench line, kernel trace + PMC passes of the real step, per-layer micro-benchmark; usage: gpu_profile.sh <tag>
set -o pipefail
R=$PWD
O=$R/gpurun_out/${1:-r3p}
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 600 $O/bench.json
cd /tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline --no-infer"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- $BENCH > $O/kt.log 2>&1; echo "kt rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmcM -o runc -- $BENCH > $O/pmcM.log 2>&1; echo "pmcM rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF -o runc -- $BENCH > $O/pmcF.log 2>&1; echo "pmcF rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW -o runc -- $BENCH > $O/pmcW.log 2>&1; echo "pmcW rc=$?"
cd $R
python scripts/analyze_trace.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/kernel_summary.txt 2>&1
python scripts/pmc_mfma.py $O/pmcM $O/pmc_mfma.txt > /dev/null 2>&1; echo "mfma rc=$?"
python scripts/pmc_traffic.py $O/pmcF $O/pmcW $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; echo "traffic rc=$?"
timeout -k 10 300 python scripts/bench_kernels.py > $O/bench_kernels.txt 2>&1; echo "bk rc=$?"
# keep the merge-back small: the raw traces are large
find $O/kt $O/pmcM $O/pmcF $O/pmcW -name "*.csv" -size +20M -delete 2>/dev/null
du -sh $O
