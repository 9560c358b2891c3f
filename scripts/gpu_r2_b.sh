#!/bin/bash
# Round-2 second GPU pass: whole GPU suite (new tests included, the no-alloc test last), MFMA counter pass, DP traces.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2b
mkdir -p $OUT
cd $R
echo "pytest" > $OUT/progress.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
echo "no_alloc alone" >> $OUT/progress.txt
timeout -k 10 240 python -u -m pytest tests/test_zz_no_alloc.py -x -v --timeout 120 --timeout-method thread > $OUT/no_alloc.log 2>&1 || { tail -30 $OUT/no_alloc.log; exit 1; }
tail -1 $OUT/no_alloc.log
echo "bench" >> $OUT/progress.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/b_vitb.json 2> $OUT/b_vitb.err || { tail -20 $OUT/b_vitb.err; exit 1; }
cut -c1-300 $OUT/b_vitb.json; grep -c AccumulateGrad $OUT/b_vitb.err
echo "bench native comm one rank" >> $OUT/progress.txt
MMAE_DP_SINGLE_RANK=1 MMAE_COMM=native timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/b_native.json 2> $OUT/b_native.err || { tail -20 $OUT/b_native.err; exit 1; }
cut -c1-300 $OUT/b_native.json
MMAE_DP_SINGLE_RANK=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/b_torchdist.json 2> $OUT/b_torchdist.err || { tail -20 $OUT/b_torchdist.err; exit 1; }
cut -c1-300 $OUT/b_torchdist.json
echo "pmc mfma" >> $OUT/progress.txt
cd /tmp; export TMPDIR=/tmp
MMAE_SIDE_STREAM=0 MMAE_DEC_STREAMS=0 timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/mfma.log 2>&1 || { tail -20 $OUT/mfma.log; exit 1; }
python3 $R/scripts/pmc_mfma_table.py $(find $OUT/mfma -name '*counter_collection.csv' | head -1) $OUT/pmc_mfma.json > $OUT/pmc_mfma_table.txt
rm -rf $OUT/mfma
head -30 $OUT/pmc_mfma_table.txt
cd $R
echo "dp trace (no start-up broadcast: the order of events of the earlier stalls)" >> $OUT/progress.txt
DP_TRACE_NO_BCAST=1 timeout -k 10 300 python3 scripts/dp_stall_trace.py 5 $OUT/dp_stall_nobcast > $OUT/dp_trace_nobcast.log 2>&1
grep attempt $OUT/dp_trace_nobcast.log
echo "dp trace" >> $OUT/progress.txt
timeout -k 10 200 python3 scripts/dp_stall_trace.py 3 $OUT/dp_stall > $OUT/dp_trace.log 2>&1
grep attempt $OUT/dp_trace.log
echo "done" >> $OUT/progress.txt
