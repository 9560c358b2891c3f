"""Experiment: N=768 GEMMs as ping-pong kernel on the first 85 row-tiles + 128x128 kernel on the remaining rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
from scripts.bench_gemm import timeit, r

if __name__ == "__main__":
    M, N = 25344, 768
    L = ops._L()
    for K, epi in [(768, 0), (768, 2), (2304, 0), (3072, 0), (3072, 2)]:
        a, w = r(M, K), r(N, K)
        bias = torch.randn(N, device="cuda")
        aux = r(M, N, dt=torch.float32) if epi == 2 else None
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
        L.mmae_gemm_nt_set_variant(1)
        t1 = timeit(lambda: ops.gemm_nt(epi, a, w, bias, aux=aux, out=out))
        for split in (21760, 16384 + 4096):
            def both():
                L.mmae_gemm_nt_set_variant(5)
                ops.gemm_nt(epi, a[:split], w, bias, aux=None if aux is None else aux[:split], out=out[:split])
                L.mmae_gemm_nt_set_variant(1)
                ops.gemm_nt(epi, a[split:], w, bias, aux=None if aux is None else aux[split:], out=out[split:])
            t2 = timeit(both)
            print(f"K={K} epi={epi}: t128 {t1*1e6:7.1f} us | pp[:{split}] + t128 tail {t2*1e6:7.1f} us")
        L.mmae_gemm_nt_set_variant(0)
