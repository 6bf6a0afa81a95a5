#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r2b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "wgrad_group" > $O/pytest_group.log 2>&1; echo "group rc=$?"; tail -15 $O/pytest_group.log
timeout -k 10 200 python scripts/bench_kernels.py --only br. --modes wgrad > $O/bk_br.txt 2>&1; echo "bk rc=$?"; cat $O/bk_br.txt
timeout -k 10 900 python -m pytest tests/test_round2_api_gpu.py tests/test_configs_gpu.py tests/test_model_gpu.py tests/test_train_entry_gpu.py -m gpu -q > $O/pytest_rest.log 2>&1; echo "rest rc=$?"; tail -25 $O/pytest_rest.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-infer > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 400 $O/bench.json
