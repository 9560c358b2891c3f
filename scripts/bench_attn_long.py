"""Attention at the ViT-L/16 384^2 shapes (encoder 289 x 289 at head_dim 64; decoder 576 x 289 and 576 x 576 at head_dim 32,
256 images): forward with the keys resident / streamed, backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
from scripts.bench_gemm import timeit, r

if __name__ == "__main__":
    L = ops._L()
    for name, B, H, nq, nk, d in [("L enc", 256, 16, 289, 289, 64), ("L dec cross", 256, 8, 576, 289, 32), ("L dec self", 256, 8, 576, 576, 32),
                                  ("B enc", 256, 12, 99, 99, 64), ("B dec self", 256, 8, 196, 196, 32)]:
        if nq == nk:
            qkv = r(B * nq, 3 * H * d); q, k_, v = qkv[:, :H*d], qkv[:, H*d:2*H*d], qkv[:, 2*H*d:]
            dqkv = torch.empty_like(qkv); dq, dk, dv = dqkv[:, :H*d], dqkv[:, H*d:2*H*d], dqkv[:, 2*H*d:]
        else:
            q = r(B * nq, H * d); kv = r(B * nk, 2 * H * d); k_, v = kv[:, :H*d], kv[:, H*d:]
            dq = torch.empty_like(q); dkv = torch.empty_like(kv); dk, dv = dkv[:, :H*d], dkv[:, H*d:]
        fl = 4.0 * B * H * nq * nk * d
        res = []
        for variant in (1, 2, 3, 0):
            L.mmae_attn_fwd_set_variant(variant)
            L.mmae_attn_fwd_set_one_pass(2)
            t = timeit(lambda: ops.attn_fwd(q, k_, v, B, H, nq, nk, d, d ** -0.5))
            L.mmae_attn_fwd_set_one_pass(1)
            t1 = timeit(lambda: ops.attn_fwd(q, k_, v, B, H, nq, nk, d, d ** -0.5))
            L.mmae_attn_fwd_set_one_pass(0)
            res.append(f"{t*1e6:6.1f} / {t1*1e6:6.1f} us {fl/t/1e12:5.0f} / {fl/t1/1e12:5.0f} TF")
        L.mmae_attn_fwd_set_variant(0)
        o, lse = ops.attn_fwd(q, k_, v, B, H, nq, nk, d, d ** -0.5)
        do = r(B * nq, H * d)
        bres = []
        for variant in (1, 3, 0):
            L.mmae_attn_bwd_set_variant(variant)
            t2 = timeit(lambda: ops.attn_bwd(q, k_, v, o, do, lse, dq, dk, dv, B, H, nq, nk, d, d ** -0.5))
            bres.append(f"{t2*1e6:7.1f} us {2*fl/t2/1e12:6.1f} TF")
        L.mmae_attn_bwd_set_variant(0)
        print(f"attn {name:12s}: fwd resident {res[0]} | streamed {res[1]} | swizzled {res[2]} | auto {res[3]} || bwd padded / swizzled / auto {' | '.join(bres)}")
