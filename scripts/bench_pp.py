"""Times one gemm_nt variant at a few large shapes: python scripts/bench_pp.py [variant]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
from scripts.bench_gemm import timeit, r  # noqa

if __name__ == "__main__":
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    ops._L().mmae_gemm_nt_set_variant(variant)
    for name, m, n, k, epi in [("qkv", 25344, 2304, 768, 0), ("fc1f32", 25344, 3072, 768, 4), ("sq4k", 4096, 4096, 4096, 0),
                               ("big", 8192, 8192, 2048, 0), ("n768k3072", 25344, 768, 3072, 0), ("1round", 21760, 768, 768, 0)]:
        a, w = r(m, k), r(n, k)
        t = timeit(lambda: ops.gemm_nt(epi, a, w, None))
        print(f"{name:10s} M={m} N={n} K={k}: {t*1e6:8.1f} us {2*m*n*k/t/1e12:7.1f} TF")
