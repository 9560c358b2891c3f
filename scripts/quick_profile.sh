#!/bin/bash
# One serial (single-stream) kernel trace of the default bench and its per-kernel table: gpurun_out/<name>/kernel_table.txt
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-quick}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
MMAE_SIDE_STREAM=0 MMAE_DEC_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o s -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/serial.log 2>&1
ST=$(find $OUT/serial -name '*kernel_trace.csv' | head -1)
python3 $R/scripts/trace_table.py $ST 8 > $OUT/kernel_table.txt
rm -rf $OUT/serial
head -60 $OUT/kernel_table.txt
