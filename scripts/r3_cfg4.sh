#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3_cfg4
mkdir -p $OUT
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/t.log 2>&1 || { tail -60 $OUT/t.log; exit 1; }
tail -2 $OUT/t.log
timeout -k 10 300 python bench.py --config vitb_rgbd_semseg --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
python3 -c "
import json;d=json.load(open('$OUT/b.json'));print(d['value'], d['ms_per_step'], d['peak_device_memory_gb'])
r=d['roofline']
for k,v in r['kernels'].items(): print(k, v)
for k,v in r['families'].items(): print(k, v)
"
