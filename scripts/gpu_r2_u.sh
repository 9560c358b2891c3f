#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2u
mkdir -p $OUT
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm" --timeout 120 --timeout-method thread > $OUT/t.log 2>&1 || { tail -30 $OUT/t.log; exit 1; }
tail -1 $OUT/t.log
timeout -k 10 200 python scripts/bench_gemm.py tn > $OUT/tn.log 2>&1 || { tail -20 $OUT/tn.log; exit 1; }
grep -v amdgpu.ids $OUT/tn.log
