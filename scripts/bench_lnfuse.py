"""Residual product + LayerNorm: the two kernels one after the other against mmae_gemm_nt_resid_ln (idle CUs of the tail
launch normalise the rows that are already complete), at the encoder's two shapes, operands rotated through more than the
Infinity Cache.  MMAE_LN_TAIL_SHARE (read once per process) sets the share of the full-tile rows handed to the idle CUs:
run once per value.  Also times the depth standardisation kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
from scripts.bench_gemm import timeit, r

if __name__ == "__main__":
    for name, M, N, K in (("proj", 25344, 768, 768), ("fc2", 25344, 768, 3072), ("L proj", 18496 * 4, 1024, 1024)):
        copies = 6
        As = [r(M, K) for _ in range(copies)]
        w, bias = r(N, K) * 0.1, torch.randn(N, device="cuda")
        res = [r(M, N, dt=torch.float32) for _ in range(copies)]
        g, b = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
        it = [0]
        def two():
            i = it[0] % copies; it[0] += 1
            out = ops.gemm_nt(ops.EPI_RESID, As[i], w, bias, aux=res[i])
            ops.layernorm_fwd(out, g, b, 1e-6)
        def fused():
            i = it[0] % copies; it[0] += 1
            ops.gemm_nt_resid_ln(As[i], w, bias, res[i], g, b, 1e-6)
        def gemm_only():
            i = it[0] % copies; it[0] += 1
            ops.gemm_nt(ops.EPI_RESID, As[i], w, bias, aux=res[i])
        t2, tf, tg = timeit(two, 60), timeit(fused, 60), timeit(gemm_only, 60)
        print(f"{name:8s} M={M} N={N} K={K}: product {tg*1e6:6.1f} us, + LayerNorm {t2*1e6:6.1f} us, fused call {tf*1e6:6.1f} us "
              f"(share {os.environ.get('MMAE_LN_TAIL_SHARE', 'default')})")
    d = torch.rand(256, 1, 224, 224, device="cuda")
    print(f"standardize_depth 256 x 224^2: {timeit(lambda: ops.standardize_depth(d), 50)*1e6:.1f} us")
