#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2w
mkdir -p $OUT
cd $R
timeout -k 10 200 python scripts/bench_gemm.py nt > $OUT/nt.log 2>&1 || { tail -20 $OUT/nt.log; exit 1; }
grep -v amdgpu.ids $OUT/nt.log | cut -c1-60,96-400
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/b_$i.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/b_$i.json'));print(d['value'], d['ms_per_step'])"
done
