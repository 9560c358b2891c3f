#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2z
mkdir -p $OUT
cd $R
for L in hip ra5 ra7; do
  echo "lib $L" >> $OUT/pp.log
  BENCH_LIB=multimae_amd/libmmae_$L.so timeout -k 10 120 python scripts/bench_gemm.py pp 2>&1 | grep "res\|qkv " >> $OUT/pp.log || { tail -20 $OUT/pp.log; exit 1; }
done
cat $OUT/pp.log
