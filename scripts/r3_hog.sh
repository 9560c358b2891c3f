#!/bin/bash
# The default bench beside a stand-in for a collective (scripts/hog: H workgroups resident for busy_us every period_us on a
# stream of their own), with the one-workgroup-per-CU kernels planned for all CUs and for (CUs - H): "HOG=H,busy,period BUDGET"
# per argument, "-" for no hog
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3_hog
mkdir -p $OUT
cd $R
for rep in 1 2; do
for cfg in "$@"; do
  hog=$(echo $cfg | cut -d: -f1); budget=$(echo $cfg | cut -d: -f2)
  envs="MMAE_CU_BUDGET=$budget"
  if [ "$hog" != "-" ]; then envs="$envs HOG=$hog"; fi
  env $envs timeout -k 10 300 python scripts/hog/bench_with_lib.py multimae_amd/libmmae_hip.so --steps 30 --warmup 8 --no-cpu-baseline --no-roofline > $OUT/b.json 2> $OUT/b.err || { tail -20 $OUT/b.err; exit 1; }
  python3 -c "import json;d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]);print('$cfg', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done
done
