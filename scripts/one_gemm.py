"""Runs one kernel shape a few times (for rocprofv3 --pmc passes): python3 scripts/one_gemm.py {nt|ntk|tn|attn_bwd|attn_fwd} [variant]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "nt"
if len(sys.argv) > 2:
    ops._L().mmae_gemm_nt_set_variant(int(sys.argv[2]))
M = 25344
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
if which == "nt":      # qkv: plain bf16 epilogue, K=768
    a, w, b = r(M, 768), r(2304, 768), torch.randn(2304, device="cuda")
    fn = lambda: ops.gemm_nt(0, a, w, b)
elif which == "ntk":   # long K
    a, w = r(M, 3072), r(768, 3072)
    fn = lambda: ops.gemm_nt(0, a, w, None)
elif which == "tn":
    x, y = r(M, 3072), r(M, 768)
    out = torch.zeros(3072, 768, device="cuda")
    fn = lambda: ops.gemm_tn(x, y, out=out, accumulate=True)
else:
    B, H, n, d = 256, 12, 99, 64
    qkv = r(B * n, 3 * H * d); q, k, v = qkv[:, :H*d], qkv[:, H*d:2*H*d], qkv[:, 2*H*d:]
    dqkv = torch.empty_like(qkv)
    o, lse = ops.attn_fwd(q, k, v, B, H, n, n, d, d ** -0.5)
    do = r(B * n, H * d)
    if which == "attn_fwd":
        fn = lambda: ops.attn_fwd(q, k, v, B, H, n, n, d, d ** -0.5)
    else:
        fn = lambda: ops.attn_bwd(q, k, v, o, do, lse, dqkv[:, :H*d], dqkv[:, H*d:2*H*d], dqkv[:, 2*H*d:], B, H, n, n, d, d ** -0.5)
for _ in range(5):
    fn()
torch.cuda.synchronize()
