#!/bin/bash
# One serial (single-stream) kernel trace of the default bench reduced to the kernel sequence of its last step: gpurun_out/<name>/sequence.txt
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-seq}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
MMAE_SIDE_STREAM=0 MMAE_DEC_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/serial -o s -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/serial.log 2>&1
ST=$(find $OUT/serial -name '*kernel_trace.csv' | head -1)
python3 $R/scripts/trace_sequence.py $ST > $OUT/sequence.txt
python3 $R/scripts/trace_table.py $ST 4 > $OUT/kernel_table.txt
rm -rf $OUT/serial
