#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2g
mkdir -p $OUT
cd $R
echo "gemm tests" > $OUT/progress.txt
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm_nt" --timeout 120 --timeout-method thread > $OUT/nt_test.log 2>&1 || { tail -30 $OUT/nt_test.log; exit 1; }
tail -2 $OUT/nt_test.log
echo "gemm bench" >> $OUT/progress.txt
timeout -k 10 300 python3 scripts/bench_gemm.py nt > $OUT/nt_bench.log 2>&1 || { tail -20 $OUT/nt_bench.log; exit 1; }
cat $OUT/nt_bench.log
echo "done" >> $OUT/progress.txt
echo "ln bench" >> $OUT/progress.txt
for R in 0 1 2 4; do
  echo "MMAE_LN_FWD_ROWS=$R" >> $OUT/ln_bench.log
  MMAE_LN_FWD_ROWS=$R timeout -k 10 120 python3 scripts/bench_ln.py >> $OUT/ln_bench.log 2>&1 || { tail -20 $OUT/ln_bench.log; exit 1; }
done
cut -c1-75 $OUT/ln_bench.log
timeout -k 10 200 python -u -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "layernorm" --timeout 120 --timeout-method thread > $OUT/ln_test.log 2>&1 || { tail -30 $OUT/ln_test.log; exit 1; }
tail -1 $OUT/ln_test.log
echo "done2" >> $OUT/progress.txt
