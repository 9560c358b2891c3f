#!/bin/bash
# In-step effect of the library's tuning switches: one bench run per setting, same box.
# usage: sweep_env.sh "VAR=a" "VAR=b" ...   (the default configuration is run first and last)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sweep
mkdir -p $OUT
cd $R
run() {
  env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/b.json'));print('$*', d['value'], d['ms_per_step'])"
}
run MMAE_NOP=1
for setting in "$@"; do run $setting; done
run MMAE_NOP=2
