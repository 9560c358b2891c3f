"""Per-kernel parity of the HIP library (through the C ABI) against fp32 PyTorch math / the oracle.
Run on the GPU box: python -m pytest tests -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu

BF16_TOL = 2e-2   # relative, bf16 bar of the parity contract
FP32_TOL = 1e-4   # relative, fp32 bar


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from multimae_amd import ops as _ops
    _ops._L()
    return _ops


def _randn(*shape, seed=0, dtype=torch.bfloat16, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to("cuda").to(dtype)


# ---------------------------------------------------------------------------------------------- GEMM
@pytest.fixture(params=[1, 4, 5, 6, 8, 7], ids=["tile128", "persist128", "pingpong256", "pingpong256_full_tiles_only", "pingpong256_half_tiles_only",
                                              "two_wg_128x256"])
def nt_variant(request, ops):
    ops._L().mmae_gemm_nt_set_variant(5 if request.param in (6, 8) else request.param)
    ops._L().mmae_gemm_nt_set_half_tiles({6: 0, 8: 2}.get(request.param, 1))  # default: 128- or 64-row tail tiles by cost
    yield request.param
    ops._L().mmae_gemm_nt_set_variant(0)
    ops._L().mmae_gemm_nt_set_half_tiles(1)


# the last three: several rounds of 256x256 tiles per CU (tiles dealt from the per-XCD counters) at 4, 6 and 12 k-tiles per tile
@pytest.mark.parametrize("M,N,K", [(396, 2304, 768), (25344, 768, 768), (1000, 256, 256), (128, 3072, 768), (777, 768, 3072), (130, 2128, 256),
                                   (50176, 1024, 256), (33000, 1280, 384), (25344, 2304, 768)])
def test_gemm_nt_bias(ops, nt_variant, M, N, K):
    a, w = _randn(M, K, seed=1), _randn(N, K, seed=2, scale=0.05)
    bias = _randn(N, seed=3, dtype=torch.float32)
    out = ops.gemm_nt(ops.EPI_BF16, a, w, bias)
    ref = a.float() @ w.float().t() + bias
    assert out.dtype == torch.bfloat16 and rel_err(out.float(), ref) < 4e-3
    out32 = ops.gemm_nt(ops.EPI_F32, a, w, bias)
    assert rel_err(out32, ref) < FP32_TOL
    nob = ops.gemm_nt(ops.EPI_F32, a, w, None)
    assert rel_err(nob, a.float() @ w.float().t()) < FP32_TOL


def test_gemm_nt_asymmetric_identity(ops, nt_variant):
    # A = I picks out rows of W^T: catches a transposed / permuted fragment map that random data averages out
    K = 128
    a = torch.eye(K, device="cuda", dtype=torch.bfloat16)
    w = (torch.arange(256 * K, device="cuda").reshape(256, K) % 251).to(torch.bfloat16)
    out = ops.gemm_nt(ops.EPI_F32, a, w, None)
    assert torch.equal(out, w.float().t().contiguous())


def test_gemm_nt_epilogues(ops, nt_variant):
    M, N, K = 1188, 1024, 256
    a, w = _randn(M, K, seed=4), _randn(N, K, seed=5, scale=0.08)
    bias = _randn(N, seed=6, dtype=torch.float32)
    lin = a.float() @ w.float().t() + bias
    h, act = ops.gemm_nt(ops.EPI_GELU, a, w, bias)
    assert rel_err(h.float(), lin) < 4e-3
    assert rel_err(act.float(), F.gelu(h.float())) < 4e-3
    resid = _randn(M, N, seed=7, dtype=torch.float32)
    out = ops.gemm_nt(ops.EPI_RESID, a, w, bias, aux=resid)
    assert out.dtype == torch.float32 and rel_err(out, resid + lin) < FP32_TOL
    hsave = _randn(M, N, seed=8)
    dg = ops.gemm_nt(ops.EPI_DGELU, a, w, None, aux=hsave)
    x = hsave.float().requires_grad_(True)
    F.gelu(x).backward(a.float() @ w.float().t())
    assert rel_err(dg.float(), x.grad) < 4e-3


@pytest.mark.parametrize("M,N,K", [(25344, 768, 768), (25344, 768, 3072), (18496, 1024, 1024), (4000, 768, 768), (50176, 256, 1024),
                                   (300, 768, 256), (1188, 512, 768)])
def test_gemm_nt_resid_ln_equals_the_two_kernels(ops, M, N, K):
    """mmae_gemm_nt_resid_ln = the residual product followed by the LayerNorm of its output.  At the encoder shapes (a tail
    launch that leaves CUs idle) the leading rows are normalised by workgroups of the tail launch, the rest by the
    LayerNorm kernel: the same per-row arithmetic, so everything is bitwise what the two separate calls give."""
    a, w = _randn(M, K, seed=31), _randn(N, K, seed=32, scale=0.05)
    bias = _randn(N, seed=33, dtype=torch.float32)
    resid = _randn(M, N, seed=34, dtype=torch.float32) * 3.0
    gamma, beta = _randn(N, seed=35, dtype=torch.float32) + 1.0, _randn(N, seed=36, dtype=torch.float32)
    out0 = ops.gemm_nt(ops.EPI_RESID, a, w, bias, aux=resid)
    ln0, mean0, rstd0 = ops.layernorm_fwd(out0, gamma, beta, 1e-6)
    for _ in range(2):
        out, ln, mean, rstd = ops.gemm_nt_resid_ln(a, w, bias, resid, gamma, beta, 1e-6)
        assert torch.equal(out, out0) and torch.equal(mean, mean0) and torch.equal(rstd, rstd0)
        assert torch.equal(ln.view(torch.int16), ln0.view(torch.int16))
    ref = torch.nn.functional.layer_norm(resid + a.float() @ w.float().t() + bias, (N,), gamma, beta, 1e-6)
    assert rel_err(ln.float(), ref) < 4e-3
    # without a bias, and with the residual read through a row stride
    wide = torch.zeros(M, N + 64, dtype=torch.float32, device="cuda")
    wide[:, 64:] = resid
    out1, ln1, _, _ = ops.gemm_nt_resid_ln(a, w, None, wide[:, 64:], gamma, beta, 1e-6)
    out2 = ops.gemm_nt(ops.EPI_RESID, a, w, None, aux=resid)
    assert torch.equal(out1, out2) and torch.equal(ln1.view(torch.int16), ops.layernorm_fwd(out2, gamma, beta, 1e-6)[0].view(torch.int16))


def test_gemm_nt_strided_views(ops, nt_variant):
    # operands / outputs that are column slices of wider buffers (how qkv and kv are consumed)
    M, K = 400, 256
    big = _randn(M, 3 * K, seed=9)
    a = big[:, K:2 * K]
    w = _randn(512, K, seed=10, scale=0.1)
    outbuf = torch.zeros(M, 1024, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(ops.EPI_BF16, a, w, None, out=outbuf[:, 512:])
    assert rel_err(outbuf[:, 512:].float(), a.float() @ w.float().t()) < 4e-3
    assert outbuf[:, :512].abs().max().item() == 0


def test_gemm_nt_epilogues_on_column_slices(ops, nt_variant):
    """Every epilogue with its output, second output and epilogue input as column slices of wider buffers (row strides
    different from N, two full 256-row panels and a ragged one), and the columns outside the slices left untouched."""
    M, N, K = 700, 512, 256
    a, w = _randn(M, K, seed=21), _randn(N, K, seed=22, scale=0.08)
    bias = _randn(N, seed=23, dtype=torch.float32)
    lin = a.float() @ w.float().t() + bias
    resid_big = _randn(M, N + 256, seed=24, dtype=torch.float32)
    out_big = torch.full((M, 2 * N), 7.0, dtype=torch.float32, device="cuda")
    ops.gemm_nt(ops.EPI_RESID, a, w, bias, aux=resid_big[:, 256:], out=out_big[:, N:])
    assert rel_err(out_big[:, N:], resid_big[:, 256:] + lin) < FP32_TOL
    assert (out_big[:, :N] == 7.0).all()
    h_big = _randn(M, N + 64, seed=25)
    dg_big = torch.full((M, N + 128), 3.0, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(ops.EPI_DGELU, a, w, None, aux=h_big[:, 64:], out=dg_big[:, :N])
    x = h_big[:, 64:].float().requires_grad_(True)
    F.gelu(x).backward(a.float() @ w.float().t())
    assert rel_err(dg_big[:, :N].float(), x.grad) < 4e-3
    assert (dg_big[:, N:] == 3.0).all()
    f32_big = torch.full((M, N + 8), -1.0, dtype=torch.float32, device="cuda")
    ops.gemm_nt(ops.EPI_F32, a, w, bias, out=f32_big[:, 8:])
    assert rel_err(f32_big[:, 8:], lin) < FP32_TOL and (f32_big[:, :8] == -1.0).all()


@pytest.fixture(params=[1, 2], ids=["tile128", "pingpong256"])
def tn_variant(request, ops):
    ops._L().mmae_gemm_tn_set_variant(request.param)
    yield request.param
    ops._L().mmae_gemm_tn_set_variant(0)


@pytest.mark.parametrize("M,N,K", [(25344, 768, 768), (396, 2304, 768), (1000, 256, 1024), (50176, 256, 256), (200, 768, 1088), (64, 8, 8),
                                   (25344, 2304, 768), (3000, 768, 3072), (130, 512, 256)])
def test_gemm_tn(ops, tn_variant, M, N, K):
    x, y = _randn(M, N, seed=11), _randn(M, K, seed=12)
    out = ops.gemm_tn(x, y)
    ref = x.float().t() @ y.float()
    assert out.shape == (N, K) and rel_err(out, ref) < FP32_TOL
    out2 = ops.gemm_tn(x, y, out=out.clone(), accumulate=True)
    assert rel_err(out2, 2 * ref) < FP32_TOL
    again, cs = ops.gemm_tn(x, y, colsum_out=True)
    assert torch.equal(again, out), "split-M reduction must be bitwise reproducible"
    assert rel_err(cs, x.float().sum(0)) < FP32_TOL, "fused column sums (bias gradient)"
    acc = cs.clone()
    ops.gemm_tn(x, y, out=out2, accumulate=True, colsum_out=acc, colsum_accumulate=True)
    assert rel_err(acc, 2 * x.float().sum(0)) < FP32_TOL and rel_err(out2, 3 * ref) < FP32_TOL


@pytest.mark.parametrize("budget", [248, 200, 96])
def test_gemm_with_a_cu_budget(ops, budget):
    """mmae_set_cu_budget: the one-workgroup-per-CU kernels plan their tiles, tails and splits for fewer CUs (left free for a
    collective's kernels); same results to rounding (the split of the weight-gradient reduction changes), at the step's shapes
    with a tail launch (N = 768), without (N = 2304) and for the split-M weight gradient."""
    L = ops._L()
    a, w, bias = _randn(25344, 768, seed=61), _randn(768, 768, seed=62, scale=0.05), _randn(768, seed=63, dtype=torch.float32)
    w2 = _randn(2304, 768, seed=64, scale=0.05)
    res = torch.randn(25344, 768, device="cuda")
    dy = _randn(25344, 2304, seed=65)
    want = (ops.gemm_nt(ops.EPI_BF16, a, w, bias), ops.gemm_nt(ops.EPI_RESID, a, w, bias, aux=res), ops.gemm_nt(ops.EPI_BF16, a, w2, None),
            ops.gemm_tn(dy, a))
    try:
        assert L.mmae_set_cu_budget(budget) == budget
        got = (ops.gemm_nt(ops.EPI_BF16, a, w, bias), ops.gemm_nt(ops.EPI_RESID, a, w, bias, aux=res), ops.gemm_nt(ops.EPI_BF16, a, w2, None),
               ops.gemm_tn(dy, a))
    finally:
        assert L.mmae_set_cu_budget(0) >= 256
    for g, r in zip(got[:3], want[:3]):
        assert torch.equal(g, r)               # a product's accumulation order does not depend on the tile plan
    assert rel_err(got[3], want[3]) < 1e-6     # the reduction over rows is split differently
    assert L.mmae_set_cu_budget(8) >= 256 and L.mmae_set_cu_budget(0) >= 256   # below 32: ignored


def test_gemm_random_ragged_shapes(ops, nt_variant):
    """A seeded sweep of small and ragged problems (any M, N and K multiples of 8 that fill no tile evenly) through every
    forward-product variant and, once, through both weight-gradient variants: bias and GELU epilogues against fp32 torch."""
    rng = np.random.default_rng(20)
    for i in range(16):
        M, N, K = int(rng.integers(1, 1500)), 8 * int(rng.integers(1, 90)), 8 * int(rng.integers(1, 60))
        a, w = _randn(M, K, seed=100 + i), _randn(N, K, seed=200 + i, scale=0.05)
        bias = _randn(N, seed=300 + i, dtype=torch.float32)
        lin = a.float() @ w.float().t() + bias
        assert rel_err(ops.gemm_nt(ops.EPI_F32, a, w, bias), lin) < FP32_TOL, (M, N, K)
        h, act = ops.gemm_nt(ops.EPI_GELU, a, w, bias)
        assert rel_err(h.float(), lin) < 4e-3 and rel_err(act.float(), F.gelu(h.float())) < 4e-3, (M, N, K)
        if nt_variant == 1:
            for v in (1, 2):
                ops._L().mmae_gemm_tn_set_variant(v)
                try:
                    y = _randn(M, N, seed=400 + i)
                    dw, cs = ops.gemm_tn(y, a, colsum_out=True)
                finally:
                    ops._L().mmae_gemm_tn_set_variant(0)
                assert rel_err(dw, y.float().t() @ a.float()) < FP32_TOL and rel_err(cs, y.float().sum(0)) < FP32_TOL, (M, N, K, v)


def test_gemm_tn_asymmetric(ops):
    M = 256
    x = torch.eye(M, device="cuda", dtype=torch.bfloat16)[:, :128].contiguous()      # [M,128]: picks rows 0..127 of y
    y = (torch.arange(M * 136, device="cuda").reshape(M, 136) % 239).to(torch.bfloat16)
    out = ops.gemm_tn(x, y)
    assert torch.equal(out, y[:128].float())


def test_gemm_tn_pingpong_asymmetric_and_views(ops, tn_variant):
    # X = rows of the identity: out[n] = Y[n] exactly; operands are column slices of wider buffers
    M, N, K = 600, 512, 256
    xb = torch.zeros(M, 1024, device="cuda", dtype=torch.bfloat16)
    xb[:N, 256:256 + N] = torch.eye(N, device="cuda", dtype=torch.bfloat16)
    yb = (torch.arange(M * 768, device="cuda").reshape(M, 768) % 241).to(torch.bfloat16)
    x, y = xb[:, 256:256 + N], yb[:, 512:512 + K]
    out, cs = ops.gemm_tn(x, y, colsum_out=True)
    assert torch.equal(out, y[:N].float()) and torch.equal(cs, torch.ones(N, device="cuda"))


def test_colsum(ops):
    x = _randn(25344, 3072, seed=13)
    assert rel_err(ops.colsum(x), x.float().sum(0)) < FP32_TOL
    x2 = _randn(396, 768, seed=14)[:, 256:512]
    assert rel_err(ops.colsum(x2), x2.float().sum(0)) < FP32_TOL


# ---------------------------------------------------------------------------------------------- LayerNorm
@pytest.fixture(params=[0, 1], ids=["rows", "generic"])
def ln_bwd_variant(request, ops):
    ops._L().mmae_layernorm_bwd_set_variant(request.param)
    yield request.param
    ops._L().mmae_layernorm_bwd_set_variant(0)


@pytest.mark.parametrize("rows,dim", [(25344, 768), (25343, 768), (397, 256), (50176, 256), (50, 1024), (7, 64), (1, 256)])
def test_layernorm(ops, rows, dim, ln_bwd_variant):
    x = _randn(rows, dim, seed=15, dtype=torch.float32) * 2 + 0.3
    g = _randn(dim, seed=16, dtype=torch.float32) * 0.1 + 1
    b = _randn(dim, seed=17, dtype=torch.float32) * 0.1
    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    xr = x.clone().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (dim,), gr, br, 1e-6)
    assert rel_err(y.float(), ref) < 4e-3
    assert rel_err(mean, x.mean(1)) < 1e-5 and rel_err(rstd, 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)) < 1e-5
    dy = _randn(rows, dim, seed=18)
    dres = _randn(rows, dim, seed=19, dtype=torch.float32)
    ref.backward(dy.float())
    dx, dxb, dg, db = ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, want_bf16=True)
    assert rel_err(dx, xr.grad + dres) < FP32_TOL
    assert rel_err(dxb.float(), dx) < 4e-3
    assert rel_err(dg, gr.grad) < FP32_TOL and rel_err(db, br.grad) < FP32_TOL
    dx0, none, _, _ = ops.layernorm_bwd(dy, x, g, mean, rstd)
    assert none is None and rel_err(dx0, xr.grad) < FP32_TOL


# ---------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, scale):
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * scale
    p = s.softmax(-1)
    return torch.einsum("bhqk,bhkd->bhqd", p, v), torch.logsumexp(s, -1)


@pytest.fixture(params=[1, 2, 3], ids=["two_kernels", "fused", "two_kernels_swizzled"])
def attn_bwd_variant(request, ops):
    ops._L().mmae_attn_bwd_set_variant(request.param)
    yield request.param
    ops._L().mmae_attn_bwd_set_variant(0)


@pytest.mark.parametrize("B,H,Nq,Nk,d", [(3, 12, 99, 99, 64), (2, 8, 196, 196, 32), (2, 8, 196, 99, 32), (1, 2, 17, 17, 32),
                                          (2, 3, 128, 128, 64), (1, 4, 33, 97, 64), (2, 8, 197, 197, 32), (1, 2, 289, 289, 64), (1, 3, 300, 40, 32),
                                          (1, 2, 250, 250, 64), (1, 2, 520, 520, 32), (2, 8, 576, 289, 32), (1, 3, 290, 305, 64),
                                          (1, 1, 1, 1, 32), (1, 1, 1, 50, 64), (2, 1, 40, 1, 32), (3, 2, 2, 3, 64)])  # a single query / key
def test_attention_fwd_bwd(ops, attn_bwd_variant, B, H, Nq, Nk, d):
    scale = d ** -0.5
    # self-attention reads q,k,v out of a packed [B*N, 3*H*d] buffer; cross-attention q from [B*Nq, H*d], k/v from [B*Nk, 2*H*d]
    if Nq == Nk:
        qkv = _randn(B * Nq, 3 * H * d, seed=20)
        qv, kv_, vv = qkv[:, :H * d], qkv[:, H * d:2 * H * d], qkv[:, 2 * H * d:]
    else:
        qbuf = _randn(B * Nq, H * d, seed=21)
        kvbuf = _randn(B * Nk, 2 * H * d, seed=22)
        qv, kv_, vv = qbuf, kvbuf[:, :H * d], kvbuf[:, H * d:]
    o, lse = ops.attn_fwd(qv, kv_, vv, B, H, Nq, Nk, d, scale)

    def heads(t, n):
        return t.float().reshape(B, n, H, d).permute(0, 2, 1, 3).clone().requires_grad_(True)
    q4, k4, v4 = heads(qv, Nq), heads(kv_, Nk), heads(vv, Nk)
    oref, lref = _attn_ref(q4, k4, v4, scale)
    assert rel_err(o.float().reshape(B, Nq, H, d).permute(0, 2, 1, 3), oref) < 6e-3
    assert rel_err(lse, lref) < 1e-4

    d_o = _randn(B * Nq, H * d, seed=23)
    oref.backward(d_o.float().reshape(B, Nq, H, d).permute(0, 2, 1, 3))
    if Nq == Nk:
        dqkv = torch.zeros_like(qkv)
        dq, dk, dv = dqkv[:, :H * d], dqkv[:, H * d:2 * H * d], dqkv[:, 2 * H * d:]
    else:
        dq = torch.zeros_like(qbuf); dkv = torch.zeros_like(kvbuf)
        dk, dv = dkv[:, :H * d], dkv[:, H * d:]
    ops.attn_bwd(qv, kv_, vv, o, d_o, lse, dq, dk, dv, B, H, Nq, Nk, d, scale)

    def back(t4, n):
        return t4.grad.permute(0, 2, 1, 3).reshape(B * n, H * d)
    assert rel_err(dq.float(), back(q4, Nq)) < BF16_TOL
    assert rel_err(dk.float(), back(k4, Nk)) < BF16_TOL
    assert rel_err(dv.float(), back(v4, Nk)) < BF16_TOL


@pytest.mark.parametrize("B,H,Nq,Nk,d", [(2, 16, 289, 289, 64), (1, 2, 520, 520, 32), (2, 8, 576, 289, 32), (3, 12, 99, 99, 64), (1, 2, 17, 17, 32),
                                          (1, 4, 33, 97, 64), (1, 3, 300, 40, 32), (1, 2, 1153, 1153, 64), (1, 1, 1, 1, 32), (1, 1, 1, 50, 64),
                                          (2, 1, 40, 1, 32), (2, 3, 128, 128, 64), (1, 2, 65, 64, 64), (1, 2, 64, 65, 32), (1, 1, 200, 161, 64)])
@pytest.mark.parametrize("variant", [2, 3, 10, 23])
def test_attention_forward_with_streamed_keys(ops, B, H, Nq, Nk, d, variant):
    """The other forms of the forward, forced for every shape, against the fp32 reference and against the padded two-pass
    resident kernel: 2 = keys streamed through LDS in 64-key tiles with a running softmax (the default where a head's K / V
    do not fit the LDS: any key length in one call); 3 = resident K / V in unpadded swizzled images, one pass (the default
    where only that lets two workgroups share a CU, e.g. ViT-L/16 at 384^2: 289 encoder tokens at head_dim 64); 10 = padded
    images, one pass; 23 = swizzled images, two passes.  Tails of 1..63 keys, query blocks that do not fill their workgroup."""
    scale = d ** -0.5
    if Nq == Nk:
        qkv = _randn(B * Nq, 3 * H * d, seed=40)
        qv, kv_, vv = qkv[:, :H * d], qkv[:, H * d:2 * H * d], qkv[:, 2 * H * d:]
    else:
        qv = _randn(B * Nq, H * d, seed=41)
        kvbuf = _randn(B * Nk, 2 * H * d, seed=42)
        kv_, vv = kvbuf[:, :H * d], kvbuf[:, H * d:]
    L = ops._L()
    try:
        L.mmae_attn_fwd_set_variant(variant % 10)
        L.mmae_attn_fwd_set_one_pass(variant // 10)
        o, lse = ops.attn_fwd_single(qv, kv_, vv, B, H, Nq, Nk, d, scale)
        o2, lse2 = ops.attn_fwd_single(qv, kv_, vv, B, H, Nq, Nk, d, scale)
    finally:
        L.mmae_attn_fwd_set_variant(0)
        L.mmae_attn_fwd_set_one_pass(0)
    assert torch.equal(o.view(torch.int16), o2.view(torch.int16)) and torch.equal(lse, lse2)  # reproducible
    f = lambda t, n: t.float().reshape(B, n, H, d).permute(0, 2, 1, 3)
    oref, lref = _attn_ref(f(qv, Nq), f(kv_, Nk), f(vv, Nk), scale)
    assert rel_err(f(o, Nq), oref) < 6e-3 and rel_err(lse, lref) < 1e-4
    if Nk <= L.mmae_attn_max_len(d, 0):
        L.mmae_attn_fwd_set_variant(1)
        try:
            o1, lse1 = ops.attn_fwd_single(qv, kv_, vv, B, H, Nq, Nk, d, scale)
        finally:
            L.mmae_attn_fwd_set_variant(0)
        assert rel_err(o.float(), o1.float()) < 6e-3 and rel_err(lse, lse1) < 1e-5


@pytest.mark.parametrize("variant", [2, 3, 10])
@pytest.mark.parametrize("d", [32, 64])
def test_attention_forward_one_pass_follows_a_growing_maximum(ops, d, variant):
    """One-pass forms: the reference point of the exponentials has to move when later keys score far above the earlier ones
    (here the keys grow 6x along the sequence and the scores reach +-60), and the sums so far are rescaled each time."""
    B, H, N = 2, 3, 289
    qkv = _randn(B * N, 3 * H * d, seed=50).float()
    grow = (1.0 + 5.0 * torch.arange(N, device="cuda") / N).repeat(B)[:, None]
    qkv[:, H * d:2 * H * d] *= grow * 1.5
    qkv[:, :H * d] *= 2.0
    qkv = qkv.to(torch.bfloat16)
    q, k, v = qkv[:, :H * d], qkv[:, H * d:2 * H * d], qkv[:, 2 * H * d:]
    L = ops._L()
    try:
        L.mmae_attn_fwd_set_variant(variant % 10)
        L.mmae_attn_fwd_set_one_pass(1)
        o, lse = ops.attn_fwd_single(q, k, v, B, H, N, N, d, d ** -0.5)
    finally:
        L.mmae_attn_fwd_set_variant(0)
        L.mmae_attn_fwd_set_one_pass(0)
    f = lambda t: t.float().reshape(B, N, H, d).permute(0, 2, 1, 3)
    oref, lref = _attn_ref(f(q), f(k), f(v), d ** -0.5)
    assert lref.abs().max() > 20  # the case is what it claims to be
    assert rel_err(f(o), oref) < 6e-3 and rel_err(lse, lref) < 1e-4


def test_attention_random_shapes_every_kernel_choice(ops):
    """A seeded sweep of ragged problems (query and key counts 1..700 that fill no tile evenly, both head sizes) through every
    forward form (by size, padded resident, streamed, swizzled resident; one and two passes) and every backward form that takes
    the shape (by size, two kernels padded, one kernel, two kernels swizzled), against the fp32 reference."""
    rng = np.random.default_rng(7)
    L = ops._L()
    for case in range(20):
        d = int(rng.choice([32, 64]))
        B, H = int(rng.integers(1, 3)), int(rng.integers(1, 4))
        Nq, Nk = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        if case % 4 == 0:
            Nk = Nq
        scale = d ** -0.5
        qbuf, kvbuf = _randn(B * Nq, H * d, seed=500 + case), _randn(B * Nk, 2 * H * d, seed=600 + case)
        kv_, vv = kvbuf[:, :H * d], kvbuf[:, H * d:]
        d_o = _randn(B * Nq, H * d, seed=700 + case)

        def heads(t, n):
            return t.float().reshape(B, n, H, d).permute(0, 2, 1, 3).clone().requires_grad_(True)
        q4, k4, v4 = heads(qbuf, Nq), heads(kv_, Nk), heads(vv, Nk)
        oref, lref = _attn_ref(q4, k4, v4, scale)
        oref.backward(d_o.float().reshape(B, Nq, H, d).permute(0, 2, 1, 3))
        back = lambda t4, n: t4.grad.permute(0, 2, 1, 3).reshape(B * n, H * d)
        fits_fwd = Nk <= L.mmae_attn_max_len(d, 0)
        fits_bwd = max(Nq, Nk) <= L.mmae_attn_max_len(d, 1)
        o = lse = None
        try:
            for variant in (0, 1, 2, 3):
                if variant == 1 and not fits_fwd:
                    continue
                for passes in (1, 2):
                    L.mmae_attn_fwd_set_variant(variant)
                    L.mmae_attn_fwd_set_one_pass(passes)
                    o, lse = ops.attn_fwd(qbuf, kv_, vv, B, H, Nq, Nk, d, scale)
                    tag = (case, B, H, Nq, Nk, d, variant, passes)
                    assert rel_err(o.float().reshape(B, Nq, H, d).permute(0, 2, 1, 3), oref) < 6e-3, tag
                    assert rel_err(lse, lref) < 1e-4, tag
            L.mmae_attn_fwd_set_variant(0)
            L.mmae_attn_fwd_set_one_pass(0)
            o, lse = ops.attn_fwd(qbuf, kv_, vv, B, H, Nq, Nk, d, scale)
            for variant in (0, 1, 2, 3):
                if variant == 2 and not fits_bwd:
                    continue   # the one-kernel form needs Q, dO, K, V of a head in LDS together; ops.attn_bwd chunks the others
                L.mmae_attn_bwd_set_variant(variant)
                dq = torch.zeros_like(qbuf); dkv = torch.zeros_like(kvbuf)
                ops.attn_bwd(qbuf, kv_, vv, o, d_o, lse, dq, dkv[:, :H * d], dkv[:, H * d:], B, H, Nq, Nk, d, scale)
                tag = (case, B, H, Nq, Nk, d, "bwd", variant)
                assert rel_err(dq.float(), back(q4, Nq)) < BF16_TOL, tag
                assert rel_err(dkv[:, :H * d].float(), back(k4, Nk)) < BF16_TOL, tag
                assert rel_err(dkv[:, H * d:].float(), back(v4, Nk)) < BF16_TOL, tag
        finally:
            L.mmae_attn_fwd_set_variant(0)
            L.mmae_attn_fwd_set_one_pass(0)
            L.mmae_attn_bwd_set_variant(0)


def test_attention_length_limit(ops):
    """The keys (forward, dQ) or queries (dK/dV) of one head live in LDS: the longest sequences that fit 160 KiB run in one
    call and agree with the reference, one 32-row tile more returns MMAE_ESHAPE before anything is launched; the limits are
    what mmae_attn_max_len reports."""
    from multimae_amd import _lib
    lib = _lib.load()
    assert [lib.mmae_attn_max_len(d, b) for d in (64, 32) for b in (0, 1)] == [544, 544, 1024, 960]
    assert lib.mmae_attn_max_len(48, 0) == 0
    for d, n_ok in ((64, 544), (32, 960)):
        B, H = 1, 2
        qkv = _randn(B * n_ok, 3 * H * d, seed=30)
        q, k, v = qkv[:, :H * d], qkv[:, H * d:2 * H * d], qkv[:, 2 * H * d:]
        o, lse = ops.attn_fwd_single(q, k, v, B, H, n_ok, n_ok, d, d ** -0.5)
        f = lambda t, n=n_ok: t.float().reshape(B, n, H, d).permute(0, 2, 1, 3)
        oref, lref = _attn_ref(f(q), f(k), f(v), d ** -0.5)
        assert rel_err(f(o), oref) < 6e-3 and rel_err(lse, lref) < 1e-4
        dqkv = torch.zeros_like(qkv)
        ops.attn_bwd_single(q, k, v, o, _randn(B * n_ok, H * d, seed=31), lse, dqkv[:, :H * d], dqkv[:, H * d:2 * H * d], dqkv[:, 2 * H * d:],
                            B, H, n_ok, n_ok, d, d ** -0.5)
        assert torch.isfinite(dqkv.float()).all() and dqkv.float().abs().sum() > 0
        n_bad = (1153 if d == 64 else 1025)  # ViT-L/16 384^2 without masking at head_dim 64; one row past 1024 at head_dim 32
        big = _randn(B * n_bad, 3 * H * d, seed=32)
        qb, kb, vb = big[:, :H * d], big[:, H * d:2 * H * d], big[:, 2 * H * d:]
        lib.mmae_attn_fwd_set_variant(1)   # the resident-key forward; the default streams keys that do not fit and has no limit
        try:
            with pytest.raises(_lib.MmaeError, match="LDS"):
                ops.attn_fwd_single(qb, kb, vb, B, H, n_bad, n_bad, d, d ** -0.5)
        finally:
            lib.mmae_attn_fwd_set_variant(0)
        ob, lb = torch.zeros(B * n_bad, H * d, dtype=torch.bfloat16, device="cuda"), torch.zeros(B, H, n_bad, device="cuda")
        with pytest.raises(_lib.MmaeError, match="LDS"):
            ops.attn_bwd_single(qb, kb, vb, ob, ob, lb, torch.zeros_like(qb), torch.zeros_like(kb), torch.zeros_like(vb), B, H, n_bad, n_bad, d,
                                d ** -0.5)
    # just past the limit at head_dim 64 (576 rows, the next multiple of 32 after 544)
    qkv = _randn(576, 3 * 64, seed=33)
    lib.mmae_attn_fwd_set_variant(1)
    try:
        with pytest.raises(_lib.MmaeError, match="LDS"):
            ops.attn_fwd_single(qkv[:, :64], qkv[:, 64:128], qkv[:, 128:], 1, 1, 576, 576, 64, 0.125)
    finally:
        lib.mmae_attn_fwd_set_variant(0)


@pytest.mark.parametrize("B,H,Nq,Nk,d", [(2, 4, 1153, 1153, 64),   # ViT-L/16 at 384^2 without masking: 2 x 576 tokens + 1
                                         (1, 2, 700, 1300, 32),    # cross-attention: only the keys are chunked
                                         (1, 2, 1000, 99, 32)])    # backward: only the queries are chunked
def test_attention_beyond_one_call_runs_in_chunks(ops, B, H, Nq, Nk, d):
    """Sequences longer than the LDS-resident kernels take: the forward streams the keys (one call), the backward runs as chunks
    of queries x keys with summed partial gradients; same tolerances as the single call."""
    scale = d ** -0.5
    if Nq == Nk:
        qkv = _randn(B * Nq, 3 * H * d, seed=40)
        qv, kv_, vv = qkv[:, :H * d], qkv[:, H * d:2 * H * d], qkv[:, 2 * H * d:]
        dqkv = torch.zeros_like(qkv)
        dq, dk, dv = dqkv[:, :H * d], dqkv[:, H * d:2 * H * d], dqkv[:, 2 * H * d:]
    else:
        qv, kvbuf = _randn(B * Nq, H * d, seed=41), _randn(B * Nk, 2 * H * d, seed=42)
        kv_, vv = kvbuf[:, :H * d], kvbuf[:, H * d:]
        dq, dkv = torch.zeros_like(qv), torch.zeros_like(kvbuf)
        dk, dv = dkv[:, :H * d], dkv[:, H * d:]
    o, lse = ops.attn_fwd(qv, kv_, vv, B, H, Nq, Nk, d, scale)

    def heads(t, n):
        return t.float().reshape(B, n, H, d).permute(0, 2, 1, 3).clone().requires_grad_(True)
    q4, k4, v4 = heads(qv, Nq), heads(kv_, Nk), heads(vv, Nk)
    oref, lref = _attn_ref(q4, k4, v4, scale)
    assert rel_err(o.float().reshape(B, Nq, H, d).permute(0, 2, 1, 3), oref) < 6e-3
    assert rel_err(lse, lref) < 1e-4
    d_o = _randn(B * Nq, H * d, seed=43)
    oref.backward(d_o.float().reshape(B, Nq, H, d).permute(0, 2, 1, 3))
    ops.attn_bwd(qv, kv_, vv, o, d_o, lse, dq, dk, dv, B, H, Nq, Nk, d, scale)
    back = lambda t4, n: t4.grad.permute(0, 2, 1, 3).reshape(B * n, H * d)
    assert rel_err(dq.float(), back(q4, Nq)) < BF16_TOL
    assert rel_err(dk.float(), back(k4, Nk)) < BF16_TOL
    assert rel_err(dv.float(), back(v4, Nk)) < BF16_TOL


def test_attention_spiked_scores(ops):
    # one key dominates each row (large logits): softmax must not overflow and the padded key tile must stay masked
    B, H, N, d = 1, 2, 99, 64
    qkv = _randn(B * N, 3 * H * d, seed=24)
    qkv[:, H * d:H * d + 8] *= 30
    q, k, v = qkv[:, :H * d], qkv[:, H * d:2 * H * d], qkv[:, 2 * H * d:]
    o, lse = ops.attn_fwd(q, k, v, B, H, N, N, d, d ** -0.5)
    f = lambda t: t.float().reshape(B, N, H, d).permute(0, 2, 1, 3)
    oref, lref = _attn_ref(f(q), f(k), f(v), d ** -0.5)
    assert torch.isfinite(o.float()).all() and rel_err(f(o), oref) < 1e-2 and rel_err(lse, lref) < 1e-4


# ---------------------------------------------------------------------------------------------- masking
def test_mask_golden_bit_exact(ops):
    from tests.golden import common as G
    z = G.load("masks.npz")
    for c in range(int(z["n_cases"])):
        pre = f"c{c}_"
        k = int(z[pre + "k"])
        p = torch.from_numpy(z[pre + "p"]).cuda()
        T = p.shape[1]
        noise = torch.cat([torch.from_numpy(z[pre + f"noise{i}"]) for i in range(T)], 1).cuda()
        noise_all = torch.from_numpy(z[pre + "noise_all"]).cuda()
        mask, ids_keep, ids_restore, keep32, restore32 = ops.mask_generate(p, noise, noise_all, [196] * T, k)
        assert np.array_equal(ids_keep.cpu().numpy(), z[pre + "ids_keep"].astype(np.int64)), c
        assert np.array_equal(ids_restore.cpu().numpy(), z[pre + "ids_restore"].astype(np.int64)), c
        assert np.array_equal(mask.cpu().numpy(), z[pre + "mask"].astype(np.int64)), c
        assert torch.equal(keep32.long(), ids_keep) and torch.equal(restore32.long(), ids_restore)


@pytest.mark.parametrize("B,sizes,k", [(256, [196, 196], 98), (64, [196, 196, 196], 98), (5, [16, 16], 8), (3, [576, 576], 288),
                                       (2, [196, 196], 392), (2, [196, 196], 0), (7, [100, 37, 5], 40)])
def test_mask_vs_oracle_and_properties(ops, B, sizes, k):
    from oracle import multimae_oracle as O
    g = torch.Generator().manual_seed(100 + B)
    p, noises, noise_all = O.draw_mask_randomness(B, sizes, 1.0, g)
    masks, ids_keep, ids_restore = O.generate_masks(p, noises, noise_all, k)
    mask, gk, gr, _, _ = ops.mask_generate(p.cuda(), torch.cat(noises, 1).cuda(), noise_all.cuda(), sizes, k)
    assert torch.equal(gk.cpu(), ids_keep) and torch.equal(gr.cpu(), ids_restore)
    assert torch.equal(mask.cpu(), torch.cat(masks, 1))
    total = sum(sizes)
    assert torch.equal(torch.sort(gr, 1)[0].cpu(), torch.arange(total).expand(B, -1)), "ids_restore is a permutation"
    assert torch.equal(mask.sum(1).cpu(), torch.full((B,), total - k)), "exactly k tokens are kept"
    if k > 0:
        assert torch.equal(torch.gather(gr, 1, gk).cpu(), torch.arange(k).expand(B, -1)), "restore(keep[j]) == j"


def test_mask_ties_resolve_to_lower_index(ops):
    from oracle import multimae_oracle as O
    B, sizes, k = 4, [32, 32], 16
    p = torch.full((B, 2), 0.5)
    noises = [torch.zeros(B, 32), torch.full((B, 32), 0.25)]
    noise_all = torch.zeros(B, 64)
    masks, ids_keep, ids_restore = O.generate_masks(p, noises, noise_all, k)
    mask, gk, gr, _, _ = ops.mask_generate(p.cuda(), torch.cat(noises, 1).cuda(), noise_all.cuda(), sizes, k)
    assert torch.equal(gk.cpu(), ids_keep) and torch.equal(gr.cpu(), ids_restore) and torch.equal(mask.cpu(), torch.cat(masks, 1))


def test_mask_random_shapes_vs_oracle(ops):
    """Task counts 1..8, task lengths 1..1024 that are mostly not multiples of the 64-lane wavefront, totals up to the 2,048
    limit (every register-per-lane width of the sort, runs with only padding in them), budgets from 0 to the total, and
    coarse noise with many equal keys (ties resolve to the lower index): bit-exact against the oracle."""
    from oracle import multimae_oracle as O
    rng = np.random.default_rng(7)
    for case in range(48):
        T = int(rng.integers(1, 9))
        cap = (2048 // T)
        hi = [8, 70, 260, 1024][case % 4]
        sizes = [int(rng.integers(1, min(hi, cap) + 1)) for _ in range(T)]
        total = sum(sizes)
        k = int(rng.choice([0, total, int(rng.integers(0, total + 1))], p=[0.1, 0.1, 0.8]))
        B = int(rng.integers(1, 6))
        g = torch.Generator().manual_seed(1000 + case)
        p, noises, noise_all = O.draw_mask_randomness(B, sizes, 1.0, g)
        if case % 3 == 0:  # quantised noise: many ties inside the tasks and in the second sort
            noises = [torch.floor(n * 4) / 4 for n in noises]
            noise_all = torch.floor(noise_all * 8) / 8
        masks, ids_keep, ids_restore = O.generate_masks(p, noises, noise_all, k)
        mask, gk, gr, k32, r32 = ops.mask_generate(p.cuda(), torch.cat(noises, 1).cuda(), noise_all.cuda(), sizes, k)
        ctx = (case, sizes, k, B)
        assert torch.equal(gk.cpu(), ids_keep), ctx
        assert torch.equal(gr.cpu(), ids_restore), ctx
        assert torch.equal(mask.cpu(), torch.cat(masks, 1)), ctx
        assert torch.equal(k32.long(), gk) and torch.equal(r32.long(), gr), ctx


# ---------------------------------------------------------------------------------------------- optimiser
def test_adamw_sumsq_cast(ops):
    n, n_decay = 1_000_004, 600_000
    g0 = torch.Generator().manual_seed(5)
    p = torch.randn(n, generator=g0).cuda(); grad = torch.randn(n, generator=g0).cuda()
    ref_p = [p[:n_decay].clone().requires_grad_(True), p[n_decay:].clone().requires_grad_(True)]
    opt = torch.optim.AdamW([{"params": [ref_p[0]], "weight_decay": 0.05}, {"params": [ref_p[1]], "weight_decay": 0.0}],
                            lr=1e-3, betas=(0.9, 0.95), eps=1e-8)
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    wb = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    for step in (1, 2, 3):
        ref_p[0].grad = grad[:n_decay].clone() * 0.5; ref_p[1].grad = grad[n_decay:].clone() * 0.5
        opt.step()
        ops.adamw_flat(p, grad, m, v, n_decay, 1e-3, 0.9, 0.95, 1e-8, 0.05, step, grad_scale=0.5, w_bf16=wb)
    assert rel_err(p[:n_decay], ref_p[0].detach()) < 1e-5 and rel_err(p[n_decay:], ref_p[1].detach()) < 1e-5
    assert torch.equal(wb, p.to(torch.bfloat16))
    assert abs(ops.sumsq(grad).item() - (grad.double() ** 2).sum().item()) / (grad.double() ** 2).sum().item() < 1e-5
    w = torch.randn(300, 200, generator=g0).cuda()
    d, dt = ops.cast_transpose(w)
    assert torch.equal(d, w.to(torch.bfloat16)) and torch.equal(dt, w.t().contiguous().to(torch.bfloat16))
    assert torch.equal(ops.cast_bf16(w), w.to(torch.bfloat16))


@pytest.mark.parametrize("n", [4, 1_000_004, 5_000_000])
def test_adamw_by_products_clear_the_gradients_and_leave_the_norm(ops, n):
    """mmae_adamw_flat_ex: the same update as mmae_adamw_flat bitwise, the gradients zeroed behind the read, and partial sums
    whose fixed-order total is the squared norm of the raw gradients (before grad_scale); sizes below one block, with a ragged
    last block, and beyond the 4,096-block grid."""
    g0 = torch.Generator().manual_seed(n)
    p = torch.randn(n, generator=g0).cuda(); grad = torch.randn(n, generator=g0).cuda()
    state = [torch.zeros_like(p) for _ in range(2)] + [torch.zeros_like(p) for _ in range(2)]
    p2, g2 = p.clone(), grad.clone()
    parts = torch.full((ops.adamw_parts(n) + 3,), 7.0, device="cuda")
    ops.adamw_flat(p, grad, state[0], state[1], n // 2 // 4 * 4, 1e-3, 0.9, 0.95, 1e-8, 0.05, 1, grad_scale=0.25)
    ops.adamw_flat(p2, g2, state[2], state[3], n // 2 // 4 * 4, 1e-3, 0.9, 0.95, 1e-8, 0.05, 1, grad_scale=0.25, clear_grad=True,
                   sumsq_part=parts)
    assert torch.equal(p, p2) and torch.equal(state[0], state[2]) and torch.equal(state[1], state[3])
    assert not g2.any() and torch.equal(parts[-3:], torch.full((3,), 7.0, device="cuda"))  # writes exactly adamw_parts(n) sums
    want = (grad.double() ** 2).sum().item()
    got = ops.sum_partials(parts[:-3]).item()
    assert abs(got - want) <= 1e-5 * want
    assert ops.sum_partials(parts[:-3]).item() == got
    with pytest.raises(ValueError):
        ops.adamw_flat(p2, g2, state[2], state[3], 0, 1e-3, 0.9, 0.95, 1e-8, 0.05, 2, sumsq_part=parts[:max(ops.adamw_parts(n) - 1, 0)])


# ---------------------------------------------------------------------------------------------- losses
def _loss_fns():
    from oracle import multimae_oracle as O
    return {"mse": (0, False, O.masked_mse), "l1": (1, False, O.masked_l1), "msenp": (0, True, O.masked_mse), "l1np": (1, True, O.masked_l1)}


@pytest.mark.parametrize("tagc", ["c3", "c1"])
def test_masked_loss_golden(ops, tagc):
    from tests.golden import common as G
    z = G.load("losses.npz")
    for nm, (kind, npix, _) in _loss_fns().items():
        for tag in [f"{nm}_{tagc}"] + ([f"mse_rowzero_{tagc}", f"mse_allzero_{tagc}", f"mse_nomask_{tagc}"] if nm == "mse" else []) + \
                   ([f"l1_nomask_{tagc}"] if nm == "l1" else []):
            pred = torch.from_numpy(z[tag + "_pred"]).cuda()
            target = torch.from_numpy(z[tag + "_target"]).cuda()
            mask = torch.from_numpy(z[tag + "_mask"].astype(np.int64)).cuda() if (tag + "_mask") in z else None
            loss, dpred = ops.masked_loss(kind, npix, pred, target, mask, 16)
            ref = float(z[tag + "_loss"])
            assert abs(loss.item() - ref) <= FP32_TOL * max(abs(ref), 1e-12), tag
            if (tag + "_dpred") in z:
                gref = torch.from_numpy(z[tag + "_dpred"])
                g = dpred.cpu()
                assert torch.equal(g.isnan(), gref.isnan()), tag
                ok = ~gref.isnan()
                assert rel_err(g[ok], gref[ok]) < FP32_TOL, tag
            else:
                assert dpred.abs().max().item() == 0, tag  # nothing masked: constant zero loss, zero gradient


@pytest.mark.parametrize("C", [3, 1])
def test_masked_loss_full_size_vs_oracle(ops, C):
    B = 16
    g = torch.Generator().manual_seed(7)
    pred = torch.randn(B, C, 224, 224, generator=g)
    target = torch.randn(B, C, 224, 224, generator=g)
    mask = (torch.rand(B, 196, generator=g) > 0.25).long()
    for nm, (kind, npix, fn) in _loss_fns().items():
        pr = pred.clone().requires_grad_(True)
        ref = fn(pr, target, mask=mask, patch_size=16, norm_pix=npix)
        ref.backward()
        loss, dpred = ops.masked_loss(kind, npix, pred.cuda(), target.cuda(), mask.cuda(), 16)
        assert abs(loss.item() - ref.item()) / abs(ref.item()) < FP32_TOL, nm
        assert rel_err(dpred.cpu(), pr.grad) < FP32_TOL, nm
        # bf16 predictions (what the decoders emit); the gradient comes back in bf16
        pb = pred.cuda().to(torch.bfloat16)
        lb, db = ops.masked_loss(kind, npix, pb, target.cuda(), mask.cuda(), 16)
        pr2 = pb.float().cpu().requires_grad_(True)
        ref2 = fn(pr2, target, mask=mask, patch_size=16, norm_pix=npix)
        ref2.backward()
        assert abs(lb.item() - ref2.item()) / abs(ref2.item()) < FP32_TOL and db.dtype == torch.bfloat16
        assert rel_err(db.float().cpu(), pr2.grad) < 4e-3


def test_transpose_multi_matches_torch(ops):
    torch.manual_seed(3)
    shapes = [(768, 2304), (100, 36), (64, 64), (8, 264), (3072, 768), (72, 40)]
    offs, off, rows, tiles = [], 0, [], 0
    for n, k in shapes:
        offs.append(off)
        rows.append([off, n, k, tiles])
        tiles += ((n + 63) // 64) * ((k + 63) // 64)
        off += (n * k + 7) // 8 * 8
    src = torch.randn(off, device="cuda").bfloat16()
    dst = torch.zeros_like(src)
    ops.transpose_multi(src, dst, torch.tensor(rows, dtype=torch.int32, device="cuda"), tiles)
    for (n, k), o in zip(shapes, offs):
        assert torch.equal(dst[o:o + n * k].view(k, n), src[o:o + n * k].view(n, k).t())


# ---------------------------------------------------------------------------------------------- config-4 pieces
def test_masked_ce_golden(ops):
    from tests.golden import common as G
    z = G.load("losses.npz")
    for tag, ls in (("ce", 0.0), ("ce_ls", 0.1), ("ce_nomask", 0.0)):
        logits = torch.from_numpy(z[tag + "_pred"]).cuda()
        target = torch.from_numpy(z[tag + "_target"]).cuda()
        mask = torch.from_numpy(z[tag + "_mask"].astype(np.int64)).cuda() if (tag + "_mask") in z else None
        loss, dl = ops.masked_ce(logits, target, mask, 4, ls)
        ref = float(z[tag + "_loss"])
        assert abs(loss.item() - ref) <= FP32_TOL * abs(ref), tag
        assert rel_err(dl.cpu(), z[tag + "_dpred"]) < FP32_TOL, tag


def test_masked_ce_full_size_vs_oracle(ops):
    """133 classes at 56x56 (the stride-4 class map of a 224^2 image), one sample without masked tokens, ignored pixels."""
    from oracle import multimae_oracle as O
    B, C, S = 6, 133, 56
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(B, C, S, S, generator=g) * 3
    target = torch.randint(0, C, (B, S, S), generator=g)
    target[1, :5] = -100
    mask = (torch.rand(B, 196, generator=g) > 0.25).long()
    mask[2] = 0
    for ls in (0.0, 0.1):
        lr = logits.clone().requires_grad_(True)
        ref = O.masked_ce(lr, target, mask=mask, patch_size=16, stride=4, label_smoothing=ls)
        ref.backward()
        loss, dl = ops.masked_ce(logits.cuda(), target.cuda(), mask.cuda(), 4, ls)
        assert abs(loss.item() - ref.item()) / abs(ref.item()) < FP32_TOL
        g_hip, g_ref = dl.cpu(), lr.grad
        assert torch.equal(g_hip.isnan(), g_ref.isnan()) and g_ref[2].isnan().all()
        ok = ~g_ref.isnan()
        assert rel_err(g_hip[ok], g_ref[ok]) < FP32_TOL
    # bf16 logits, bf16 gradient; nothing masked at all -> 0 and zero gradient
    lb = logits.cuda().bfloat16()
    loss_b, dl_b = ops.masked_ce(lb, target.cuda(), mask.cuda(), 4, 0.0)
    lr = lb.float().cpu().requires_grad_(True)
    ref = O.masked_ce(lr, target, mask=mask, patch_size=16, stride=4)
    ref.backward()
    assert abs(loss_b.item() - ref.item()) / abs(ref.item()) < FP32_TOL and dl_b.dtype == torch.bfloat16
    ok = ~lr.grad.isnan()
    assert rel_err(dl_b.float().cpu()[ok], lr.grad[ok]) < 4e-3
    loss0, dl0 = ops.masked_ce(logits.cuda(), target.cuda(), torch.zeros_like(mask).cuda(), 4, 0.0)
    assert loss0.item() == 0.0 and dl0.abs().max().item() == 0.0


def test_pack_patches_class_map_and_embedding_backward(ops):
    """Class-map domain next to an image domain: packed rows = embedding rows of the kept patches; the embedding
    gradient equals index_add of the row gradients, bitwise reproducible."""
    torch.manual_seed(5)
    B, nh, nw, P_img, P_cls, ncls, cdim, k = 3, 4, 4, 16, 4, 11, 8, 9
    img = torch.randn(B, 2, nh * P_img, nw * P_img, device="cuda")
    cmap = torch.randint(0, ncls, (B, nh * P_cls, nw * P_cls), device="cuda")
    emb = torch.randn(ncls, cdim, device="cuda")
    ntok = nh * nw
    keep = torch.stack([torch.randperm(2 * ntok)[:k] for _ in range(B)]).to(torch.int32).cuda()
    k_img, k_cls = 2 * P_img * P_img, cdim * P_cls * P_cls
    ktot = (k_img + k_cls + 8 + 63) // 64 * 64
    a = ops.pack_patches([img, cmap], [P_img, P_cls], keep, ktot, nh, nw, [None, emb])
    e_img = emb[cmap].permute(0, 3, 1, 2)  # [B, cdim, H, W]
    rows_img = img.reshape(B, 2, nh, P_img, nw, P_img).permute(0, 2, 4, 1, 3, 5).reshape(B, ntok, k_img)
    rows_cls = e_img.reshape(B, cdim, nh, P_cls, nw, P_cls).permute(0, 2, 4, 1, 3, 5).reshape(B, ntok, k_cls)
    want = torch.zeros(B * k, ktot, device="cuda")
    for b in range(B):
        for j in range(k):
            t, pos = divmod(int(keep[b, j]), ntok)
            if t == 0:
                want[b * k + j, :k_img] = rows_img[b, pos]
            else:
                want[b * k + j, k_img:k_img + k_cls] = rows_cls[b, pos]
            want[b * k + j, k_img + k_cls + t] = 1.0
    assert torch.equal(a.float(), want.bfloat16().float())

    d_rows = torch.randn(B * k, ktot, device="cuda").bfloat16()
    d_cls = d_rows[:, k_img:k_img + k_cls]
    got = ops.class_emb_bwd(d_cls, cmap, keep, ntok, nh, nw, P_cls, ncls, cdim)
    ref = torch.zeros(ncls, cdim, dtype=torch.float64, device="cuda")
    for b in range(B):
        for j in range(k):
            t, pos = divmod(int(keep[b, j]), ntok)
            if t != 1:
                continue
            y0, x0 = (pos // nw) * P_cls, (pos % nw) * P_cls
            cls = cmap[b, y0:y0 + P_cls, x0:x0 + P_cls].reshape(-1)
            gpatch = d_cls[b * k + j].double().reshape(cdim, P_cls * P_cls)
            ref.index_add_(0, cls, gpatch.t().contiguous())
    assert rel_err(got.cpu(), ref.cpu()) < 1e-6
    assert torch.equal(got, ops.class_emb_bwd(d_cls, cmap, keep, ntok, nh, nw, P_cls, ncls, cdim))
    acc = got.clone()
    ops.class_emb_bwd(d_cls, cmap, keep, ntok, nh, nw, P_cls, ncls, cdim, out=acc, accumulate=True)
    assert torch.allclose(acc, 2 * got)


def test_unpatchify_patch4_and_fp32_output(ops):
    torch.manual_seed(9)
    B, C, P, nh, nw, Dd = 2, 9, 4, 4, 4, 64
    x = torch.randn(B * nh * nw, Dd, device="cuda").bfloat16()
    w = (torch.randn(C * P * P, Dd, device="cuda") * 0.1).bfloat16()
    bias = torch.randn(C * P * P, device="cuda")
    ref = (x.float() @ w.float().t() + bias).reshape(B, nh, nw, C, P, P).permute(0, 3, 1, 4, 2, 5).reshape(B, C, nh * P, nw * P)
    img32 = ops.unpatchify_gemm(x, w, bias, B, C, P, nh, nw, out_f32=True)
    img16 = ops.unpatchify_gemm(x, w, bias, B, C, P, nh, nw)
    assert img32.dtype == torch.float32 and rel_err(img32.cpu(), ref.cpu()) < 1e-5
    assert img16.dtype == torch.bfloat16 and torch.equal(img16, img32.bfloat16())
    rows = ops.patchify_bf16(img32, P)  # the adjoint layout: back to one row per token
    back = (x.float() @ w.float().t() + bias)
    assert torch.equal(rows, back.bfloat16()) or rel_err(rows.float().cpu(), back.cpu()) < 4e-3
    assert torch.equal(ops.patchify_bf16(img16, P), rows)


@pytest.mark.parametrize("B,C,P,nh,nw", [(3, 3, 16, 14, 14),     # rgb at 224^2
                                         (2, 1, 16, 14, 14),     # depth
                                         (2, 133, 4, 14, 14),    # class logits at stride 4 (59.6 KB strip)
                                         (1, 3, 16, 24, 24),     # 384^2
                                         (1, 200, 4, 14, 20),    # strip beyond 64 KB: the direct form
                                         (2, 5, 8, 3, 7)])
def test_patchify_bit_exact(ops, B, C, P, nh, nw):
    """image [B,C,nh*P,nw*P] -> rows [B*nh*nw, C*P*P] in (c, ph, pw) order: a pure permutation (plus the bf16 rounding of an
    fp32 image), so bit-exact against torch for bf16 and fp32 inputs and for both kernel forms."""
    g = torch.Generator().manual_seed(B * 1000 + C)
    img = torch.randn(B, C, nh * P, nw * P, generator=g).cuda()
    ref = img.reshape(B, C, nh, P, nw, P).permute(0, 2, 4, 1, 3, 5).reshape(B * nh * nw, C * P * P).bfloat16()
    assert torch.equal(ops.patchify_bf16(img, P), ref)
    assert torch.equal(ops.patchify_bf16(img.bfloat16(), P), ref)


def test_adamw_clip_and_skip_on_device(ops):
    """Clipping = torch.nn.utils.clip_grad_norm_ followed by AdamW; a skipped update changes nothing and does not advance
    the bias-correction step (the reference skips optimizer.step(), utils/native_scaler.py:41-46)."""
    n, n_decay = 40_000, 24_000
    g0 = torch.Generator().manual_seed(8)
    p = torch.randn(n, generator=g0).cuda()
    ref = [p[:n_decay].clone().requires_grad_(True), p[n_decay:].clone().requires_grad_(True)]
    opt = torch.optim.AdamW([{"params": [ref[0]], "weight_decay": 0.05}, {"params": [ref[1]], "weight_decay": 0.0}],
                            lr=1e-3, betas=(0.9, 0.95), eps=1e-8)
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    step_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    scales = [1.0, 30.0, 0.01, 5.0]     # gradient magnitudes: the second one trips the skip threshold
    world_scale = 0.5
    for it, sc in enumerate(scales):
        grad = torch.randn(n, generator=g0).cuda() * sc
        true = grad * world_scale
        norm = true.norm().item()
        before = p.clone()
        ops.adamw_flat(p, grad, m, v, n_decay, 1e-3, 0.9, 0.95, 1e-8, 0.05, it + 1, grad_scale=world_scale,
                       grad_sumsq=ops.sumsq(grad), max_norm=150.0, skip_norm=2000.0, step_dev=step_dev)
        if norm >= 2000.0:
            assert torch.equal(p, before)
            continue
        ref[0].grad = true[:n_decay].clone(); ref[1].grad = true[n_decay:].clone()
        total = torch.nn.utils.clip_grad_norm_(ref, 150.0)
        assert abs(total.item() - norm) / norm < 1e-5
        opt.step()
    assert step_dev.item() == 3
    assert rel_err(p[:n_decay], ref[0].detach()) < 1e-5 and rel_err(p[n_decay:], ref[1].detach()) < 1e-5


def test_truncated_depth_standardisation(ops):
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    z = G.load("depth_std.npz")
    got = ops.standardize_depth(torch.from_numpy(z["depth"]).cuda())
    assert rel_err(got.cpu(), z["out"]) < FP32_TOL
    g = torch.Generator().manual_seed(17)
    d = torch.rand(6, 1, 224, 224, generator=g) * 9.0
    d[1] = torch.round(d[1])                 # heavy ties, also across the two cut ranks
    d[2] = -d[2]                             # negative values
    d[3, 0, :120] = 2.5                      # more than half the map is one value: both cuts land on it
    d[4] = torch.randn(1, 224, 224, generator=g) * 1e-3   # mixed signs, small scale
    ref = O.standardize_depth(d)
    got = ops.standardize_depth(d.cuda())
    for b in range(d.shape[0]):
        assert rel_err(got[b].cpu(), ref[b]) < FP32_TOL, b
    x = d.cuda().clone()
    ops.standardize_depth(x, out=x)          # in place
    assert torch.equal(x, got)


@pytest.mark.parametrize("h,w", [(64, 64), (37, 100), (176, 176), (256, 256), (260, 256), (384, 384)])
def test_truncated_depth_standardisation_sizes(ops, h, w):
    """Every instantiation of the register form (16 / 32 / 49 / 64 values per thread: up to 65,536 pixels, the last size exactly),
    a map that does not fill its threads evenly, and the memory form beyond (384^2 is the ViT-L configuration's size): against the
    oracle, with ties across both cut ranks in one sample, and repeatable bitwise."""
    from oracle import multimae_oracle as O
    g = torch.Generator().manual_seed(h * 1000 + w)
    d = torch.rand(3, 1, h, w, generator=g) * 5.0 - 1.0
    d[1] = torch.round(d[1] * 2) / 2
    ref = O.standardize_depth(d)
    got = ops.standardize_depth(d.cuda())
    for b in range(3):
        assert rel_err(got[b].cpu(), ref[b]) < FP32_TOL, b
    assert torch.equal(got, ops.standardize_depth(d.cuda()))
