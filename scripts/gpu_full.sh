#!/bin/bash
# Full GPU suite, then the default bench (ViT-B RGB-D) twice without the CPU baseline.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-full}
mkdir -p $OUT
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $OUT/t.log 2>&1 || { tail -40 $OUT/t.log; exit 1; }
tail -2 $OUT/t.log
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b_$i.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/b_$i.json'));print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
