#!/bin/bash
# One serial (single-stream) kernel trace of another bench configuration and its per-kernel table: gpurun_out/<name>/kernel_table.txt
# usage: quick_profile_config.sh <config> [name] [steps]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-vitl384}
OUT=$R/gpurun_out/${2:-quick_$CFG}
STEPS=${3:-3}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
MMAE_SIDE_STREAM=0 MMAE_DEC_STREAMS=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o s -- python3 $R/bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-roofline > $OUT/serial.log 2>&1
ST=$(find $OUT/serial -name '*kernel_trace.csv' | head -1)
python3 $R/scripts/trace_table.py $ST $((STEPS + 1)) > $OUT/kernel_table.txt
python3 $R/scripts/trace_sequence.py $ST > $OUT/sequence.txt
rm -rf $OUT/serial
