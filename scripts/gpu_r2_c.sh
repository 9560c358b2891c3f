#!/bin/bash
# MFMA counter pass, DP traces, then the one-rank benches through torch.distributed's RCCL and through the C ABI's driver.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2c
mkdir -p $OUT
echo "pmc mfma" > $OUT/progress.txt
cd /tmp; export TMPDIR=/tmp
MMAE_SIDE_STREAM=0 MMAE_DEC_STREAMS=0 timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/mfma.log 2>&1 || { tail -20 $OUT/mfma.log; exit 1; }
python3 $R/scripts/pmc_mfma_table.py $(find $OUT/mfma -name '*counter_collection.csv' | head -1) $OUT/pmc_mfma.json > $OUT/pmc_mfma_table.txt
rm -rf $OUT/mfma
head -24 $OUT/pmc_mfma_table.txt
cd $R
echo "dp trace (no start-up broadcast: the order of events of the earlier stalls)" >> $OUT/progress.txt
DP_TRACE_NO_BCAST=1 timeout -k 10 300 python3 scripts/dp_stall_trace.py 5 $OUT/dp_stall_nobcast > $OUT/dp_trace_nobcast.log 2>&1
grep attempt $OUT/dp_trace_nobcast.log
grep -q STALLED $OUT/dp_trace_nobcast.log && exit 0
echo "dp trace" >> $OUT/progress.txt
timeout -k 10 200 python3 scripts/dp_stall_trace.py 3 $OUT/dp_stall > $OUT/dp_trace.log 2>&1
grep attempt $OUT/dp_trace.log
grep -q STALLED $OUT/dp_trace.log && exit 0
echo "bench torch.distributed one rank" >> $OUT/progress.txt
MMAE_DP_SINGLE_RANK=1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/b_torchdist.json 2> $OUT/b_torchdist.err || { tail -20 $OUT/b_torchdist.err; exit 1; }
cut -c1-300 $OUT/b_torchdist.json
echo "bench native comm one rank" >> $OUT/progress.txt
MMAE_BENCH_VERBOSE=1 MMAE_DP_SINGLE_RANK=1 MMAE_COMM=native timeout -k 10 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/b_native.json 2> $OUT/b_native.err || { tail -40 $OUT/b_native.err; exit 1; }
cut -c1-300 $OUT/b_native.json
echo "done" >> $OUT/progress.txt
