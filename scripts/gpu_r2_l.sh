#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2l
mkdir -p $OUT
cd $R
for PM in 0 1 2 3; do
  echo "MMAE_P2_PRIO=$PM" >> $OUT/nt_bench.log
  MMAE_P2_PRIO=$PM timeout -k 10 200 python3 scripts/bench_gemm.py nt 2>&1 | grep -E "qkv  |fc1\+gelu|dgelu  |dA qkv|dec fc1|L fc1" | sed 's/t128.*pp256 full/.../; s/| auto.*//' >> $OUT/nt_bench.log || { tail -20 $OUT/nt_bench.log; exit 1; }
done
cut -c1-200 $OUT/nt_bench.log
