#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2kk
mkdir -p $OUT
cd $R
echo "default env: HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-unset}"
timeout -k 10 200 python scripts/attn_trace.py 2>&1 | grep -A1 "^enc"
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 200 python scripts/attn_trace.py 2>&1 | grep -A1 "^enc"
HIP_FORCE_DEV_KERNARG=0 timeout -k 10 200 python scripts/attn_trace.py 2>&1 | grep -A1 "^enc"
for v in 1 0 1 0; do
HIP_FORCE_DEV_KERNARG=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/b.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/b.json'));print('HIP_FORCE_DEV_KERNARG=$v', d['value'], d['ms_per_step'])"
done
