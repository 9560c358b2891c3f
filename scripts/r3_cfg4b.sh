#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_model.py -x -q -k "fp32_output or semseg" --timeout 300 --timeout-method thread 2>&1 | tail -4
bash scripts/quick_profile_config.sh vitb_rgbd_semseg quick_semseg3 3
head -32 gpurun_out/quick_semseg3/kernel_table.txt
cd $R && timeout -k 10 300 python bench.py --config vitb_rgbd_semseg --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'])"
