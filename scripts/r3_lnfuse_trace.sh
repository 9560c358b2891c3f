#!/bin/bash
# kernel durations of the residual product + LayerNorm micro-benchmark under the kernel trace, per share of rows handed to the idle CUs
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3_lnfuse_trace
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for share in 1.0 0.5; do
  MMAE_LN_TAIL_SHARE=$share timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$share -o s -- python3 $R/scripts/bench_lnfuse.py > $OUT/log_$share.txt 2>&1
  ST=$(find $OUT/t$share -name '*kernel_trace.csv' | head -1)
  echo "share $share" >> $OUT/tables.txt
  python3 $R/scripts/trace_table.py $ST 1 | grep -E "gemm_nt_pp|layernorm" >> $OUT/tables.txt
  rm -rf $OUT/t$share
done
cat $OUT/tables.txt
