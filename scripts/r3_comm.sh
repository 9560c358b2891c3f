#!/bin/bash
# Data-parallel tests, then the bench three ways on one box: plain, one-rank RCCL group through torch.distributed, one-rank
# native communicator (the collectives move nothing; what shows is the cost of the exchange's plumbing).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r3_comm}
mkdir -p $OUT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_dp.py tests/test_gpu_bench.py tests/test_zz_no_alloc.py -x -q --timeout 180 --timeout-method thread > $OUT/t.log 2>&1 || { tail -60 $OUT/t.log; exit 1; }
tail -2 $OUT/t.log
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29561
for mode in plain torch native; do
  case $mode in
    plain) extra="";;
    torch) export MMAE_DP_SINGLE_RANK=1; extra="--comm torch";;
    native) export MMAE_DP_SINGLE_RANK=1; extra="--comm native";;
  esac
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $extra > $OUT/b_$mode.json 2> $OUT/b_$mode.err || { tail -20 $OUT/b_$mode.err; exit 1; }
  python3 -c "import json;d=json.load(open('$OUT/b_$mode.json'));print('$mode', d['value'], d['ms_per_step'], d.get('comm'))"
done
