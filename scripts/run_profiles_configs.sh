#!/bin/bash
# Counter passes and per-kernel tables of the two other bench configurations (RGB + depth + semseg; ViT-L/16 384^2) into
# gpurun_out/prof_cfg/: FETCH_SIZE / WRITE_SIZE (HBM bytes per launch -> pmc_traffic_<config>.json, which bench.py reads for
# roofline.traffic of that configuration) and the MFMA counters (-> pmc_mfma_table_<config>.txt), each a run of its own with
# the program directly after "--".  Each GPU step has its own time limit; the script stops at the first failure.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_cfg
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
export MMAE_SIDE_STREAM=0 MMAE_DEC_STREAMS=0
for CFG in vitb_rgbd_semseg vitl384; do
  echo "$CFG trace" >> $OUT/progress.txt
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_$CFG -o s -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/serial_$CFG.log 2>&1
  python3 $R/scripts/trace_table.py $(find $OUT/serial_$CFG -name '*kernel_trace.csv' | head -1) 4 > $OUT/kernel_table_$CFG.txt
  rm -rf $OUT/serial_$CFG
  echo "$CFG fetch" >> $OUT/progress.txt
  timeout -s KILL 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$CFG -o f -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/fetch_$CFG.log 2>&1
  echo "$CFG write" >> $OUT/progress.txt
  timeout -s KILL 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$CFG -o w -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/write_$CFG.log 2>&1
  echo "$CFG mfma" >> $OUT/progress.txt
  timeout -s KILL 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma_$CFG -o m -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/mfma_$CFG.log 2>&1
  python3 $R/scripts/pmc_table.py $(find $OUT/fetch_$CFG -name '*counter_collection.csv' | head -1) $(find $OUT/write_$CFG -name '*counter_collection.csv' | head -1) $OUT/pmc_traffic_$CFG.json > $OUT/pmc_traffic_table_$CFG.txt
  python3 $R/scripts/pmc_mfma_table.py $(find $OUT/mfma_$CFG -name '*counter_collection.csv' | head -1) $OUT/pmc_mfma_$CFG.json > $OUT/pmc_mfma_table_$CFG.txt
  rm -rf $OUT/fetch_$CFG $OUT/write_$CFG $OUT/mfma_$CFG
done
echo done >> $OUT/progress.txt
