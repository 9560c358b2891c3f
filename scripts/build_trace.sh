#!/bin/bash
# Timing build of the library (gemm.hip with -DMMAE_PP_TRACE, attn.hip with -DMMAE_ATTN_TRACE) next to the product one: multimae_amd/libmmae_trace.so
set -e
cd "$(dirname "$0")/.."
make -s
mkdir -p build/trace
cp build/obj/*.o build/trace/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DMMAE_PP_TRACE -c multimae_amd/csrc/gemm.hip -o build/trace/gemm.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DMMAE_ATTN_TRACE -c multimae_amd/csrc/attn.hip -o build/trace/attn.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o multimae_amd/libmmae_trace.so build/trace/*.o -ldl
