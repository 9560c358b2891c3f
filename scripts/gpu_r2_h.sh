#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2h
mkdir -p $OUT
cd $R
for D in 0 2 4 6 9 13; do
  echo "MMAE_P2_DELAY_US=$D" >> $OUT/nt_bench.log
  MMAE_P2_DELAY_US=$D timeout -k 10 200 python3 scripts/bench_gemm.py nt 2>&1 | grep -E "qkv|fc1\+gelu|dgelu|dec fc1" | sed 's/t128.*pp256 full/.../' >> $OUT/nt_bench.log || { tail -20 $OUT/nt_bench.log; exit 1; }
done
cut -c1-260 $OUT/nt_bench.log
