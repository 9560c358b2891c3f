#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2s
mkdir -p $OUT
rm -f $OUT/pp.log
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm or gelu" --timeout 120 --timeout-method thread > $OUT/t.log 2>&1 || { tail -30 $OUT/t.log; exit 1; }
tail -1 $OUT/t.log
timeout -k 10 150 python scripts/pp_trace.py > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
grep -A12 "^fc1\|^dgelu" $OUT/trace.log | cut -c1-230
timeout -k 10 200 python scripts/bench_gemm.py nt > $OUT/nt.log 2>&1 || { tail -20 $OUT/nt.log; exit 1; }
grep -v amdgpu.ids $OUT/nt.log | cut -c1-60,96-400
