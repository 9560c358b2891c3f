#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2q
mkdir -p $OUT
cd $R
for SK in 0,0 2000,0 3000,0 3000,500 3000,1000 2000,1500 4000,2000; do
  echo "MMAE_PP_SKEW=$SK" >> $OUT/pp.log
  MMAE_PP_SKEW=$SK timeout -k 10 120 python scripts/bench_gemm.py pp >> $OUT/pp.log 2>&1 || { tail -20 $OUT/pp.log; exit 1; }
done
cat $OUT/pp.log
