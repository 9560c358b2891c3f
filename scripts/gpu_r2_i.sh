#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2i
mkdir -p $OUT
cd $R
timeout -k 10 300 python3 scripts/bench_gemm.py dec 2>&1 | sed 's/t128.*pp256 full/.../' > $OUT/dec_bench.log || { tail -20 $OUT/dec_bench.log; exit 1; }
cut -c1-250 $OUT/dec_bench.log
