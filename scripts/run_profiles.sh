#!/bin/bash
# Collects the round's profiles on the GPU box into gpurun_out/prof/ (copy what is to be kept into profiles/).
# Each GPU step runs under its own time limit; the script stops at the first failure.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
echo "bench" > $OUT/progress.txt
timeout -k 10 400 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench, other configurations" >> $OUT/progress.txt
timeout -k 10 400 python3 $R/bench.py --config vitb_rgbd_semseg > $OUT/bench_vitb_rgbd_semseg.json 2> $OUT/bench_semseg.err
timeout -k 10 600 python3 $R/bench.py --config vitl384 --steps 10 --warmup 3 > $OUT/bench_vitl384.json 2> $OUT/bench_vitl384.err
echo "serial trace" >> $OUT/progress.txt
MMAE_SIDE_STREAM=0 MMAE_DEC_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o s -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/serial.log 2>&1
echo "overlapped trace" >> $OUT/progress.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/overlap -o o -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/overlap.log 2>&1
echo "pmc fetch" >> $OUT/progress.txt
MMAE_SIDE_STREAM=0 MMAE_DEC_STREAMS=0 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/fetch.log 2>&1
echo "pmc write" >> $OUT/progress.txt
MMAE_SIDE_STREAM=0 MMAE_DEC_STREAMS=0 timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/write.log 2>&1
echo "pmc mfma" >> $OUT/progress.txt
MMAE_SIDE_STREAM=0 MMAE_DEC_STREAMS=0 timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/mfma.log 2>&1
echo "tables" >> $OUT/progress.txt
ST=$(find $OUT/serial -name '*kernel_trace.csv' | head -1); OT=$(find $OUT/overlap -name '*kernel_trace.csv' | head -1)
python3 $R/scripts/trace_table.py $ST 10 > $OUT/kernel_table.txt
python3 $R/scripts/trace_table.py $OT 8 > $OUT/kernel_table_overlapped.txt
python3 $R/scripts/trace_streams.py $OT 8 > $OUT/streams.txt
cp $(find $OUT/serial -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/overlap -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_overlapped.csv
python3 $R/scripts/pmc_table.py $(find $OUT/fetch -name '*counter_collection.csv' | head -1) $(find $OUT/write -name '*counter_collection.csv' | head -1) $OUT/pmc_traffic.json > $OUT/pmc_traffic_table.txt
python3 $R/scripts/pmc_mfma_table.py $(find $OUT/mfma -name '*counter_collection.csv' | head -1) $OUT/pmc_mfma.json > $OUT/pmc_mfma_table.txt
# the raw traces are large: keep the reductions only
rm -rf $OUT/serial $OUT/overlap $OUT/fetch $OUT/write $OUT/mfma
echo "done" >> $OUT/progress.txt
