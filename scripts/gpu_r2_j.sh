#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2j
mkdir -p $OUT
cd $R
echo "pytest" > $OUT/progress.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
echo "bench" >> $OUT/progress.txt
for i in 1 2; do
MMAE_EAGER_OPT=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/b_noeager_$i.json 2> $OUT/b_noeager.err || { tail -20 $OUT/b_noeager.err; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/b_eager_$i.json 2> $OUT/b_eager.err || { tail -20 $OUT/b_eager.err; exit 1; }
done
for f in $OUT/b_noeager_1.json $OUT/b_eager_1.json $OUT/b_noeager_2.json $OUT/b_eager_2.json; do echo $f; cut -c60-175 $f; done
MMAE_DP_SINGLE_RANK=1 MMAE_COMM=native timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/b_native.json 2> $OUT/b_native.err || { tail -20 $OUT/b_native.err; exit 1; }
cut -c60-175 $OUT/b_native.json
echo "done" >> $OUT/progress.txt
