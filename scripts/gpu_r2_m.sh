#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2m
mkdir -p $OUT
cd $R
echo "pytest" > $OUT/progress.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
echo "bench" >> $OUT/progress.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/b.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
cut -c60-175 $OUT/b.json
echo "done" >> $OUT/progress.txt
