#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2r
mkdir -p $OUT
cd $R
timeout -k 10 150 python scripts/pp_trace.py > $OUT/trace3.log 2>&1 || { tail -20 $OUT/trace3.log; exit 1; }
cut -c1-250 $OUT/trace3.log
