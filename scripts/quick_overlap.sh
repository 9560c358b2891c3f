#!/bin/bash
# Overlapped (product configuration) kernel trace of the default bench: per-kernel table and per-stream occupancy
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-overlap}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/o -o o -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/o.log 2>&1
OT=$(find $OUT/o -name '*kernel_trace.csv' | head -1)
python3 $R/scripts/trace_table.py $OT 8 > $OUT/kernel_table_overlapped.txt
python3 $R/scripts/trace_streams.py $OT 8 > $OUT/streams.txt
cp $OT $OUT/kernel_trace.csv; rm -rf $OUT/o
cat $OUT/streams.txt
head -30 $OUT/kernel_table_overlapped.txt
