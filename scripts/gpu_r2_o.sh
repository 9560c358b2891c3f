#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2o
mkdir -p $OUT
cd $R
for i in 1 2; do
for sc in 1 0; do
  MMAE_SHARED_ENC_CAST=$sc timeout -k 10 300 python bench.py --config vitl384 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/v_${sc}_$i.json 2> $OUT/v.err || { tail -20 $OUT/v.err; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/v_${sc}_$i.json'));print('shared cast $sc run $i', d['value'], d['ms_per_step'], d['peak_device_memory_gb'])"
done
done
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | head -30 > $OUT/smi.txt; nproc > $OUT/nproc.txt; uptime >> $OUT/nproc.txt
