#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2k
mkdir -p $OUT
cd $R
for B in 64 128 256; do
  timeout -k 10 300 python bench.py --config vitl384 --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/b_vitl_$B.json 2> $OUT/b_vitl_$B.err || { tail -20 $OUT/b_vitl_$B.err; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/b_vitl_$B.json'));print($B, d['value'], d['ms_per_step'], d['step_mfma_roofline_frac'], d['peak_device_memory_gb'])"
done
