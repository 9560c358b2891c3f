"""Separates the per-launch, per-tile and per-k-tile costs of the 256x256 ping-pong GEMM: N = 3072 (12 column tiles),
M = 16384 r / 3 so that the 256 CUs run exactly r rounds of full tiles, K swept.  time = c0 + r * (t_e + (K / 64) * t_k).
(Not N = 2048: with power-of-two row strides the epilogue bursts of a round collide on the memory channels and the fit
reports a per-tile cost three times the one the step's shapes see.)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimae_amd import _lib
if os.environ.get("PP_MODEL_LIB"):  # another build of the library (A/B on one box)
    _lib.LIB_PATH = os.path.abspath(os.environ["PP_MODEL_LIB"])
from multimae_amd import ops
from bench_gemm import timeit, r as rnd

ops._L().mmae_gemm_nt_set_variant(5)
ops._L().mmae_gemm_nt_set_half_tiles(0)
epi = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rows, ts = [], []
for rr in (3, 6, 9):
    for K in (256, 768, 1536, 3072):
        M, N = 16384 * rr // 3, 3072
        a, w, bias = rnd(M, K), rnd(N, K), torch.randn(N, device="cuda")
        aux = rnd(M, N, dt=torch.float32) if epi == 2 else (rnd(M, N) if epi == 3 else None)
        t = timeit(lambda: ops.gemm_nt(epi, a, w, None if epi == 3 else bias, aux=aux), iters=50) * 1e6
        print(f"r={rr} K={K:5d}: {t:7.1f} us  {2*M*N*K/t/1e6:7.1f} TF", flush=True)
        rows.append([1.0, rr, rr * K / 64]); ts.append(t)
c, *_ = np.linalg.lstsq(np.array(rows), np.array(ts), rcond=None)
print(f"epi {epi}: c0 = {c[0]:.1f} us per launch, t_e = {c[1]:.2f} us per tile, t_k = {c[2]:.3f} us per k-tile "
      f"(MFMA floor {256*256*64*2/ (2.5e15/256) * 1e6:.3f} us)")
