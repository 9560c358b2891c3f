#!/bin/bash
# Round-2 first GPU pass: new entry point, full GPU suite, the three bench configurations, counter list, DP stall trace.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2a
mkdir -p $OUT
cd $R
echo "wip_no_alloc" > $OUT/progress.txt
timeout -k 10 240 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > $OUT/no_alloc.log 2>&1 || { tail -30 $OUT/no_alloc.log; exit 1; }
echo "pytest" >> $OUT/progress.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
echo "bench vitb" >> $OUT/progress.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > $OUT/b_vitb.json 2> $OUT/b_vitb.err || { tail -20 $OUT/b_vitb.err; exit 1; }
echo "bench vitl" >> $OUT/progress.txt
timeout -k 10 400 python bench.py --config vitl384 --steps 5 --warmup 2 > $OUT/b_vitl.json 2> $OUT/b_vitl.err || { tail -20 $OUT/b_vitl.err; exit 1; }
echo "bench semseg" >> $OUT/progress.txt
timeout -k 10 300 python bench.py --config vitb_rgbd_semseg --steps 10 --warmup 3 > $OUT/b_sem.json 2> $OUT/b_sem.err || { tail -20 $OUT/b_sem.err; exit 1; }
cut -c1-400 $OUT/b_vitb.json $OUT/b_vitl.json $OUT/b_sem.json
echo "counters" >> $OUT/progress.txt
timeout -k 10 60 rocprofv3 -L > $OUT/counters.txt 2>&1
echo "dp trace" >> $OUT/progress.txt
timeout -k 10 200 python3 scripts/dp_stall_trace.py 3 $OUT/dp_stall > $OUT/dp_trace.log 2>&1
cat $OUT/dp_trace.log | grep attempt
echo "done" >> $OUT/progress.txt
