#!/bin/bash
# GPU suite, then the default bench under the settings given as arguments ("NAME=VALUE,NAME=VALUE" per run; "-" = none), twice each, alternating
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3_ab
mkdir -p $OUT
cd $R
if [ "$SKIP_TESTS" != "1" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $OUT/t.log 2>&1 || { tail -60 $OUT/t.log; exit 1; }
tail -2 $OUT/t.log
fi
for rep in 1 2; do
for cfg in "$@"; do
  envs=""
  if [ "$cfg" != "-" ]; then envs=$(echo $cfg | tr ',' ' '); fi
  env $envs timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $OUT/b.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/b.json'));print('$cfg', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done
done
