#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2p
mkdir -p $OUT
cd $R
timeout -k 10 200 python -u -m pytest tests/test_gpu_model.py -m gpu -x -q -k "never_waits or bucketwise or flat_storage" --timeout 120 --timeout-method thread > $OUT/t.log 2>&1 || { tail -30 $OUT/t.log; exit 1; }
tail -1 $OUT/t.log
for B in 128 256; do
for i in 1 2; do
  timeout -k 10 300 python bench.py --config vitl384 --batch $B --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/v_${B}_$i.json 2> $OUT/v.err || { grep -v "^frame" $OUT/v.err | tail -20; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/v_${B}_$i.json'));print('B=$B run $i', d['value'], d['ms_per_step'], d['peak_device_memory_gb'])"
done
done
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/b_$i.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
cut -c60-175 $OUT/b_$i.json
MMAE_MAX_STEPS_IN_FLIGHT=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/b0_$i.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
cut -c60-175 $OUT/b0_$i.json
done
