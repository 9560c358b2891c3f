#!/bin/bash
# attention kernels: tests, the ViT-L/16 384^2 shapes side by side, the bench line of that configuration
set -e -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/attn
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "attention" --timeout 180 --timeout-method thread 2>&1 | tail -5
timeout -k 10 200 python scripts/bench_attn_long.py 2>&1 | grep -v amdgpu | tee gpurun_out/attn/shapes.txt
timeout -k 10 400 python bench.py --config vitl384 --no-cpu-baseline > gpurun_out/attn/bench_vitl384.json 2> gpurun_out/attn/bench_vitl384.err
python -c "import json; j=json.load(open('gpurun_out/attn/bench_vitl384.json')); print(j['value'], j['ms_per_step'])"
