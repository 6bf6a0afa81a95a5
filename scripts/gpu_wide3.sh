#!/bin/bash
echo "--- wide"; AAU_IGEMM_WIDE=1 timeout -k 10 200 python scripts/bench_wide_fixed.py 2>&1 | grep "^K" || exit 1
echo "--- wide, no fetch no mfma"; AAU_IGEMM_WIDE=1 AAU_IGEMM_ABL=14 timeout -k 10 200 python scripts/bench_wide_fixed.py 2>&1 | grep "^K" || exit 1
echo "--- narrow"; AAU_IGEMM_WIDE=0 timeout -k 10 200 python scripts/bench_wide_fixed.py 2>&1 | grep "^K"
