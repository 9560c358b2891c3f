#!/bin/bash
# attention kernels: tests, the ViT-L/16 384^2 shapes side by side, the bench line of that configuration
set -e -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/attn
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_options.py -x -q -k "attention or dropout" --timeout 180 --timeout-method thread 2>&1 | tail -5
timeout -k 10 200 python scripts/bench_attn_long.py 2>&1 | grep -v amdgpu | tee gpurun_out/attn/shapes.txt
