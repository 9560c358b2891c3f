#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2n
mkdir -p $OUT
cd $R
run() { # name, env..., args
  name=$1; shift
  env "$@" > /dev/null 2>&1 || true
}
for variant in "a --steps 5 --warmup 2" "b --steps 10 --warmup 3" "c --steps 20 --warmup 3"; do
  set -- $variant; name=$1; shift
  timeout -k 10 300 python bench.py --config vitl384 "$@" --no-cpu-baseline --no-roofline > $OUT/v_$name.json 2> $OUT/v_$name.err || { tail -20 $OUT/v_$name.err; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/v_$name.json'));print('$name', d['value'], d['ms_per_step'], d['peak_device_memory_gb'])"
done
MMAE_EAGER_OPT=0 timeout -k 10 300 python bench.py --config vitl384 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/v_d.json 2> $OUT/v_d.err || { tail -20 $OUT/v_d.err; exit 1; }
python3 -c "import json;d=json.load(open('$OUT/v_d.json'));print('no eager', d['value'], d['ms_per_step'], d['peak_device_memory_gb'])"
