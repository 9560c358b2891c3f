#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3_lnfuse
mkdir -p $OUT
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "resid_ln or depth or gemm_nt_epilogues" --timeout 180 --timeout-method thread > $OUT/t.log 2>&1 || { tail -40 $OUT/t.log; exit 1; }
tail -2 $OUT/t.log
for share in 1.0 0.8 0.6; do
  MMAE_LN_TAIL_SHARE=$share timeout -k 10 120 python scripts/bench_lnfuse.py 2>&1 | grep -v Warn || exit 1
done
