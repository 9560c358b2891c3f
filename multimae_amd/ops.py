"""Tensor-level wrappers of the C ABI (include/mmae.h): argument checks, output allocation, stream plumbing.

Everything here runs the HIP kernels; there is no eager/CPU fallback.  torch is used for device memory and
for the current stream only."""
import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib

EPI_BF16, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_F32 = 0, 1, 2, 3, 4
BF16, F32 = torch.bfloat16, torch.float32


def _L():
    return _lib.load()


def _raw_stream(device_index: int = -1) -> int:
    """Handle of the current HIP stream of the device (the current device for -1).  torch.cuda.current_stream() builds a
    Stream object through several Python layers (about 9 us a call, several hundred calls a step); the C accessor is the
    same lookup without them."""
    if device_index < 0:
        device_index = torch._C._cuda_getDevice()
    return torch._C._cuda_getCurrentRawStream(device_index)


def _stream():
    return C.c_void_p(_raw_stream())


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _req(t: torch.Tensor, dtype, name: str, contiguous=True):
    if not t.is_cuda:
        raise _lib.MmaeError(f"{name}: expected a tensor on the GPU (the HIP kernels are the only implementation)")
    if t.dtype != dtype:
        raise _lib.MmaeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise _lib.MmaeError(f"{name}: expected a contiguous tensor")
    return t


def _rows2d(t: torch.Tensor, name: str):
    """(pointer tensor, rows, cols, row stride) of a 2-D view whose last dim is dense."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise _lib.MmaeError(f"{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


# When set to a list, every GEMM / attention launch appends (kernel family, algorithmic FLOPs, start event, end event);
# bench.py uses it to time the kernels with HIP events on the stream they run on.
PROFILE = None


_NT_KERNELS = {1: "gemm_nt_kernel<%d>", 4: "gemm_nt_persist_kernel<%d, 2, 2>", 5: "gemm_nt_pp_kernel<%d, 256>",
               7: "gemm_nt_p2_kernel<%d>"}
_TN_KERNELS = {1: "gemm_tn_kernel", 2: "gemm_tn_pp_kernel"}


def _nt_label(epi):
    return _NT_KERNELS.get(_L().mmae_gemm_nt_last_kernel(), "gemm_nt?<%d>") % epi


class _Timed:
    """HIP-event pair around one launch when PROFILE is a list:
    (family, flops, start, end, kernel name as rocprof prints it, algorithmic HBM bytes or None).
    Matrix kernels pass their algorithmic FLOPs, bandwidth-bound kernels `nbytes` (and flops 0.0)."""

    def __init__(self, name, flops, label=None, nbytes=None):
        self.name, self.flops, self.label, self.nbytes = name, flops, label, nbytes

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record()
            flops = self.flops() if callable(self.flops) else self.flops
            if flops is not None:  # None: nothing was launched
                PROFILE.append((self.name, flops, self.e0, self.e1, self.label() if callable(self.label) else (self.label or self.name),
                                self.nbytes))
        return False


_WS = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): kernels on one stream run in order, so they can share it."""
    idx = device.index if isinstance(device, torch.device) else torch.device(device).index
    if idx is None:
        idx = torch._C._cuda_getDevice()
    key = (idx, _raw_stream(idx))
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# --------------------------------------------------------------------------------------------- GEMMs
def gemm_nt(epi: int, a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
            aux: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """out[M,N] = a[M,K] @ w[N,K]^T with epilogue `epi`. Returns out (and the activation for EPI_GELU)."""
    _req(a, BF16, "gemm_nt a", contiguous=False)
    _req(w, BF16, "gemm_nt w", contiguous=False)
    M, K, lda = _rows2d(a, "gemm_nt a")
    N, K2, ldw = _rows2d(w, "gemm_nt w")
    if K != K2:
        raise _lib.MmaeError(f"gemm_nt: inner dims differ ({K} vs {K2})")
    out_dtype = F32 if epi in (EPI_RESID, EPI_F32) else BF16
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    else:
        _req(out, out_dtype, "gemm_nt out", contiguous=False)
    _, _, ldo = _rows2d(out, "gemm_nt out")
    out2 = torch.empty((M, N), dtype=BF16, device=a.device) if epi == EPI_GELU else None
    ldaux = 0
    if aux is not None:
        _req(aux, F32 if epi == EPI_RESID else BF16, "gemm_nt aux", contiguous=False)
        _, _, ldaux = _rows2d(aux, "gemm_nt aux")
    if bias is not None:
        _req(bias, F32, "gemm_nt bias")
    args = (epi, _ptr(a), lda, _ptr(w), ldw, M, N, K, _ptr(bias), _ptr(out), ldo, _ptr(out2), _ptr(aux), ldaux)
    if PROFILE is None:
        _lib.check(_L().mmae_gemm_nt(*args, _stream()), "gemm_nt")
    else:  # the 256x256 path may be two launches (full tiles, then half tiles for the last rows): time each as the kernel it is
        L = _L()
        try:
            L.mmae_gemm_nt_set_phase(1)
            full = lambda: L.mmae_gemm_nt_last_full_rows()
            with _Timed("gemm_nt", lambda: 2.0 * full() * N * K if full() > 0 else None, lambda: _nt_label(epi)):
                _lib.check(L.mmae_gemm_nt(*args, _stream()), "gemm_nt")
            if full() < M:
                L.mmae_gemm_nt_set_phase(2)
                with _Timed("gemm_nt", 2.0 * (M - full()) * N * K, lambda: "gemm_nt_pp_kernel<%d, %d>" % (epi, L.mmae_gemm_nt_last_tail_rows())):
                    _lib.check(L.mmae_gemm_nt(*args, _stream()), "gemm_nt")
        finally:
            L.mmae_gemm_nt_set_phase(0)
    return (out, out2) if epi == EPI_GELU else out


def gemm_nt_resid_ln(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], aux: torch.Tensor, gamma: torch.Tensor,
                     beta: torch.Tensor, eps: float = 1e-6):
    """out f32 [M,N] = aux + a @ w^T + bias, then LayerNorm(out): returns (out, ln bf16 [M,N], mean [M], rstd [M]).
    gemm_nt(EPI_RESID) followed by layernorm_fwd, bitwise; one C-ABI call so that CUs the product's tail launch leaves idle
    normalise the rows that are already complete."""
    if PROFILE is not None:  # per-kernel timing: the two kernels on their own
        out = gemm_nt(EPI_RESID, a, w, bias, aux=aux)
        ln, mean, rstd = layernorm_fwd(out, gamma, beta, eps)
        return out, ln, mean, rstd
    _req(a, BF16, "gemm_nt_resid_ln a", contiguous=False)
    _req(w, BF16, "gemm_nt_resid_ln w", contiguous=False)
    M, K, lda = _rows2d(a, "gemm_nt_resid_ln a")
    N, K2, ldw = _rows2d(w, "gemm_nt_resid_ln w")
    if K != K2:
        raise _lib.MmaeError(f"gemm_nt_resid_ln: inner dims differ ({K} vs {K2})")
    _req(aux, F32, "gemm_nt_resid_ln aux", contiguous=False)
    _, _, ldaux = _rows2d(aux, "gemm_nt_resid_ln aux")
    _req(gamma, F32, "gemm_nt_resid_ln gamma"); _req(beta, F32, "gemm_nt_resid_ln beta")
    if bias is not None:
        _req(bias, F32, "gemm_nt_resid_ln bias")
    out = torch.empty((M, N), dtype=F32, device=a.device)
    ln = torch.empty((M, N), dtype=BF16, device=a.device)
    mean = torch.empty((M,), dtype=F32, device=a.device)
    rstd = torch.empty((M,), dtype=F32, device=a.device)
    _lib.check(_L().mmae_gemm_nt_resid_ln(_ptr(a), lda, _ptr(w), ldw, M, N, K, _ptr(bias), _ptr(out), _ptr(aux), ldaux, _ptr(gamma),
                                          _ptr(beta), float(eps), _ptr(ln), _ptr(mean), _ptr(rstd), _stream()), "gemm_nt_resid_ln")
    return out, ln, mean, rstd


def gemm_tn(x: torch.Tensor, y: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False,
            colsum_out=None, colsum_accumulate: bool = False):
    """out[N,K] (f32) (+)= x[M,N]^T @ y[M,K]. colsum_out: True -> also return the column sums of x (f32 [N]); a tensor ->
    write (or accumulate) them there. Returns out, or (out, colsum) when column sums were requested."""
    _req(x, BF16, "gemm_tn x", contiguous=False)
    _req(y, BF16, "gemm_tn y", contiguous=False)
    M, N, ldx = _rows2d(x, "gemm_tn x")
    M2, K, ldy = _rows2d(y, "gemm_tn y")
    if M != M2:
        raise _lib.MmaeError(f"gemm_tn: row counts differ ({M} vs {M2})")
    if out is None:
        out = torch.empty((N, K), dtype=F32, device=x.device)
        accumulate = False
    else:
        _req(out, F32, "gemm_tn out")
        if out.numel() != N * K:
            raise _lib.MmaeError(f"gemm_tn: out has {out.numel()} elements, expected {N}x{K}")
    cs = None
    if colsum_out is True:
        cs = torch.empty((N,), dtype=F32, device=x.device)
        colsum_accumulate = False
    elif colsum_out is not None and colsum_out is not False:
        cs = _req(colsum_out, F32, "gemm_tn colsum_out")
        if cs.numel() != N:
            raise _lib.MmaeError(f"gemm_tn: colsum_out has {cs.numel()} elements, expected {N}")
    nbytes = _L().mmae_gemm_tn_workspace_bytes(M, N, K)
    ws = workspace(nbytes, x.device)
    args = (_ptr(x), ldx, _ptr(y), ldy, M, N, K, _ptr(out), int(accumulate), _ptr(cs), int(colsum_accumulate), _ptr(ws), ws.numel())
    if PROFILE is None:
        _lib.check(_L().mmae_gemm_tn(*args, _stream()), "gemm_tn")
    else:  # time the split GEMM and the slab reduction as the two kernels they are
        try:
            _L().mmae_gemm_tn_set_phase(1)
            with _Timed("gemm_tn", 2.0 * M * N * K, lambda: _TN_KERNELS.get(_L().mmae_gemm_tn_last_kernel(), "gemm_tn?")):
                _lib.check(_L().mmae_gemm_tn(*args, _stream()), "gemm_tn")
            _L().mmae_gemm_tn_set_phase(2)
            with _Timed("slab_reduce", 0.0, "slab_reduce_kernel"):
                _lib.check(_L().mmae_gemm_tn(*args, _stream()), "gemm_tn")
        finally:
            _L().mmae_gemm_tn_set_phase(0)
    return out if cs is None else (out, cs)


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    _req(x, BF16, "colsum x", contiguous=False)
    M, N, ldx = _rows2d(x, "colsum x")
    if out is None:
        out = torch.empty((N,), dtype=F32, device=x.device)
        accumulate = False
    ws = workspace(_L().mmae_colsum_workspace_bytes(M, N), x.device)
    _lib.check(_L().mmae_colsum_bf16(_ptr(x), ldx, M, N, _ptr(out), int(accumulate), _ptr(ws), ws.numel(), _stream()), "colsum")
    return out


# --------------------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-6):
    _req(x, F32, "layernorm x"); _req(gamma, F32, "layernorm gamma"); _req(beta, F32, "layernorm beta")
    dim = x.shape[-1]
    rows = x.numel() // dim
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    mean = torch.empty((rows,), dtype=F32, device=x.device)
    rstd = torch.empty((rows,), dtype=F32, device=x.device)
    with _Timed("layernorm_fwd", 0.0, "layernorm_fwd_rows_kernel", nbytes=rows * (6 * dim + 8) + 8 * dim):  # x in; y, mean, rstd out
        _lib.check(_L().mmae_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), rows, dim, eps,
                                           _stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                  dres: Optional[torch.Tensor] = None, want_bf16: bool = False, dgamma=None, dbeta=None, defer_params: bool = False):
    """Returns (dx f32 [= dres + LN'(dy)], dx bf16 or None, dgamma, dbeta); given dgamma / dbeta are accumulated into.
    defer_params: leave the parameter gradients as per-block partial rows in a buffer of their own, returned in place of
    dgamma (finish with layernorm_param_reduce, possibly on another stream); dbeta is then None."""
    _req(dy, BF16, "layernorm_bwd dy"); _req(x, F32, "layernorm_bwd x")
    dim = x.shape[-1]
    rows = x.numel() // dim
    if dres is not None:
        _req(dres, F32, "layernorm_bwd dres")
    dx = torch.empty(x.shape, dtype=F32, device=x.device)
    dxb = torch.empty(x.shape, dtype=BF16, device=x.device) if want_bf16 else None
    nbytes = _L().mmae_layernorm_bwd_workspace_bytes(rows, dim)
    # dy (bf16), x, mean, rstd (+ the residual gradient) in; dx (+ its bf16 copy) out
    moved = rows * (dim * (2 + 4 + (4 if dres is not None else 0) + 4 + (2 if want_bf16 else 0)) + 8)
    if defer_params:
        part = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        with _Timed("layernorm_bwd", 0.0, "layernorm_bwd_rows_kernel", nbytes=moved):
            _lib.check(_L().mmae_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dres), _ptr(dx), _ptr(dxb),
                                               None, None, 0, rows, dim, _ptr(part), part.numel(), _stream()), "layernorm_bwd")
        return dx, dxb, part, None
    acc = int(dgamma is not None)
    if dgamma is None:
        dgamma = torch.empty((dim,), dtype=F32, device=x.device)
        dbeta = torch.empty((dim,), dtype=F32, device=x.device)
    ws = workspace(nbytes, x.device)
    if PROFILE is None:
        _lib.check(_L().mmae_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dres), _ptr(dx), _ptr(dxb),
                                           _ptr(dgamma), _ptr(dbeta), acc, rows, dim, _ptr(ws), ws.numel(), _stream()), "layernorm_bwd")
    else:  # the row kernel timed on its own, then the parameter reduction as the second launch it is
        with _Timed("layernorm_bwd", 0.0, "layernorm_bwd_rows_kernel", nbytes=moved):
            _lib.check(_L().mmae_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dres), _ptr(dx), _ptr(dxb),
                                               None, None, 0, rows, dim, _ptr(ws), ws.numel(), _stream()), "layernorm_bwd")
        _lib.check(_L().mmae_layernorm_param_reduce(_ptr(ws), ws.numel(), rows, dim, _ptr(dgamma), _ptr(dbeta), acc, _stream()),
                   "layernorm_param_reduce")
    return dx, dxb, dgamma, dbeta


def layernorm_param_reduce(part: torch.Tensor, rows: int, dim: int, dgamma: torch.Tensor, dbeta: torch.Tensor, accumulate: bool):
    """Sum the per-block partial rows left by layernorm_bwd(defer_params=True) into dgamma / dbeta (on the current stream)."""
    _req(dgamma, F32, "layernorm_param_reduce dgamma"); _req(dbeta, F32, "layernorm_param_reduce dbeta")
    _lib.check(_L().mmae_layernorm_param_reduce(_ptr(part), part.numel(), rows, dim, _ptr(dgamma), _ptr(dbeta), int(accumulate),
                                                _stream()), "layernorm_param_reduce")


# --------------------------------------------------------------------------------------------- attention
def _head_view(t: torch.Tensor, name: str):
    """t is [B, L, H*d-ish] viewed through a row stride: return (ld) for a tensor whose rows are (b*L + i)."""
    if t.stride(-1) != 1:
        raise _lib.MmaeError(f"{name}: inner stride must be 1")
    return t.stride(-2)


def _seq_chunks(n: int, limit: int):
    """[start, end) ranges that cover n rows with the fewest chunks of at most `limit` rows, chunk starts on multiples of 32."""
    parts = -(-n // limit)
    size = (-(-n // parts) + 31) // 32 * 32
    return [(s, min(s + size, n)) for s in range(0, n, size)]


def _seq_rows(t: torch.Tensor, B: int, n: int, s: int, e: int) -> torch.Tensor:
    """Rows s..e of every sample of the (row-strided) view t [B*n, W] as a contiguous [B*(e-s), W] tensor."""
    return t.view(B, n, t.shape[-1])[:, s:e].contiguous().view(B * (e - s), t.shape[-1])


def attn_fwd(q, k, v, B, H, q_len, kv_len, head_dim, scale):
    """softmax(q k^T scale) v per (sample, head).  q: [B*q_len, >=H*d] view, k/v: [B*kv_len, ...] views (row-strided slices
    of packed projections).  Returns (o [B*q_len, H*d] bf16, lse [B,H,q_len] f32).
    One kernel call for any length: the keys and values of a head stay in LDS where they fit (up to
    mmae_attn_max_len(head_dim, 0) keys) and stream through it in 64-key tiles with a running softmax beyond."""
    return attn_fwd_single(q, k, v, B, H, q_len, kv_len, head_dim, scale)


def attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, q_len, kv_len, head_dim, scale):
    """Gradients of attn_fwd, written into the (row-strided) views dq / dk / dv.  One kernel call keeps the keys / values and
    the queries / output gradients of a head in LDS (attn_bwd_single); longer sequences run as one call per (chunk of
    queries, chunk of keys) with the log-sum-exp and output of the whole row, and the partial gradients are summed in fp32."""
    limit = _L().mmae_attn_max_len(head_dim, 1)
    if (q_len <= limit and kv_len <= limit) or limit <= 0:
        return attn_bwd_single(q, k, v, o, d_o, lse, dq, dk, dv, B, H, q_len, kv_len, head_dim, scale)
    W = H * head_dim
    dq32 = torch.zeros((B, q_len, W), dtype=F32, device=q.device)
    dk32 = torch.zeros((B, kv_len, W), dtype=F32, device=q.device)
    dv32 = torch.zeros((B, kv_len, W), dtype=F32, device=q.device)
    k_chunks = [(s, e, _seq_rows(k, B, kv_len, s, e), _seq_rows(v, B, kv_len, s, e)) for s, e in _seq_chunks(kv_len, limit)]
    mean_key = k.view(B, kv_len, H, head_dim).float().mean(dim=1).contiguous()  # [B, H, d] over the whole row
    for a0, a1 in _seq_chunks(q_len, limit):
        qa, oa, doa = (_seq_rows(t, B, q_len, a0, a1) for t in (q, o, d_o))
        lse_a = lse[:, :, a0:a1].contiguous()
        for c0, c1, kc, vc in k_chunks:
            dqa, dkc, dvc = torch.empty_like(qa), torch.empty_like(kc), torch.empty_like(vc)
            attn_bwd_single(qa, kc, vc, oa, doa, lse_a, dqa, dkc, dvc, B, H, a1 - a0, c1 - c0, head_dim, scale, mean_key=mean_key)
            dq32[:, a0:a1] += dqa.view(B, a1 - a0, W)
            dk32[:, c0:c1] += dkc.view(B, c1 - c0, W)
            dv32[:, c0:c1] += dvc.view(B, c1 - c0, W)
    dq.view(B, q_len, W).copy_(dq32)
    dk.view(B, kv_len, W).copy_(dk32)
    dv.view(B, kv_len, W).copy_(dv32)


def attn_fwd_single(q, k, v, B, H, q_len, kv_len, head_dim, scale):
    """One mmae_attn_fwd call; arguments and result as attn_fwd."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _req(t, BF16, f"attn_fwd {n}", contiguous=False)
    o = torch.empty((B * q_len, H * head_dim), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, q_len), dtype=F32, device=q.device)
    with _Timed("attn_fwd", 4.0 * B * H * q_len * kv_len * head_dim):
        _lib.check(_L().mmae_attn_fwd(_ptr(q), _head_view(q, "q"), _ptr(k), _head_view(k, "k"), _ptr(v), _head_view(v, "v"),
                                      _ptr(o), o.stride(0), _ptr(lse), B, H, q_len, kv_len, head_dim, float(scale), _stream()),
                   "attn_fwd")
    return o, lse


def attn_bwd_single(q, k, v, o, d_o, lse, dq, dk, dv, B, H, q_len, kv_len, head_dim, scale, mean_key=None):
    """One mmae_attn_bwd call (q_len and kv_len up to mmae_attn_max_len(head_dim, 1)); arguments as attn_bwd.
    mean_key [B,H,d] f32 (mmae_attn_bwd_chunk): the mean over all keys of the row when k / v are only a chunk of it."""
    if mean_key is not None:
        _req(mean_key, F32, "attn_bwd mean_key")
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (d_o, "dO"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _req(t, BF16, f"attn_bwd {n}", contiguous=False)
    _req(lse, F32, "attn_bwd lse")
    ws = workspace(_L().mmae_attn_bwd_workspace_bytes(B, H, q_len), q.device)
    with _Timed("attn_bwd", 8.0 * B * H * q_len * kv_len * head_dim):
        _lib.check(_L().mmae_attn_bwd_chunk(_ptr(q), _head_view(q, "q"), _ptr(k), _head_view(k, "k"), _ptr(v), _head_view(v, "v"),
                                            _ptr(o), _head_view(o, "o"), _ptr(d_o), _head_view(d_o, "dO"), _ptr(lse), _ptr(mean_key),
                                            _ptr(dq), _head_view(dq, "dq"), _ptr(dk), _head_view(dk, "dk"), _ptr(dv), _head_view(dv, "dv"),
                                            B, H, q_len, kv_len, head_dim, float(scale), _ptr(ws), ws.numel(), _stream()), "attn_bwd")


# --------------------------------------------------------------------------------------------- masking
def mask_generate(task_dist: torch.Tensor, noise: torch.Tensor, noise_all: torch.Tensor, tokens_per_task: Sequence[int],
                  num_encoded: int):
    """Returns (mask [B,total] i64, ids_keep [B,k] i64, ids_restore [B,total] i64, ids_keep i32, ids_restore i32)."""
    _req(task_dist, F32, "mask task_dist"); _req(noise, F32, "mask noise"); _req(noise_all, F32, "mask noise_all")
    B, T = task_dist.shape
    total = int(sum(tokens_per_task))
    if noise.shape != (B, total) or noise_all.shape != (B, total) or T != len(tokens_per_task):
        raise _lib.MmaeError("mask_generate: noise tensors must be [B, sum(tokens_per_task)]")
    dev = task_dist.device
    ids_keep = torch.empty((B, num_encoded), dtype=torch.int64, device=dev)
    ids_restore = torch.empty((B, total), dtype=torch.int64, device=dev)
    mask = torch.empty((B, total), dtype=torch.int64, device=dev)
    keep32 = torch.empty((B, num_encoded), dtype=torch.int32, device=dev)
    restore32 = torch.empty((B, total), dtype=torch.int32, device=dev)
    toks = (C.c_int * T)(*[int(n) for n in tokens_per_task])
    # two noise rows and the Dirichlet sample in; int64 ids_keep / ids_restore / mask and the int32 copies out
    moved = B * (4 * (2 * total + T) + 8 * (num_encoded + 2 * total) + 4 * (num_encoded + total))
    with _Timed("mask_generate", 0.0, "mask_generate_kernel", nbytes=moved):
        _lib.check(_L().mmae_mask_generate(_ptr(task_dist), _ptr(noise), _ptr(noise_all), toks, T, B, int(num_encoded),
                                           _ptr(ids_keep), _ptr(ids_restore), _ptr(mask), _ptr(keep32), _ptr(restore32), _stream()),
                   "mask_generate")
    return mask, ids_keep, ids_restore, keep32, restore32


# --------------------------------------------------------------------------------------------- token plumbing
def _ptr_array(ts: Sequence[Optional[torch.Tensor]]):
    return (C.c_void_p * len(ts))(*[0 if t is None else t.data_ptr() for t in ts])


def pack_patches(images: Sequence[torch.Tensor], patches: Sequence[int], ids_keep32: torch.Tensor, ktot: int, nh: int, nw: int,
                 class_embs: Optional[Sequence[Optional[torch.Tensor]]] = None):
    """images[t]: f32 [B,C_t,nh*P_t,nw*P_t], or an int64 class map [B,nh*P_t,nw*P_t] when class_embs[t] (f32 [classes, C_t]) is given."""
    class_embs = list(class_embs) if class_embs is not None else [None] * len(images)
    B = images[0].shape[0]
    chans = []
    for im, P, ce in zip(images, patches, class_embs):
        if ce is None:
            _req(im, F32, "pack_patches image")
            want = (B, im.shape[1], nh * P, nw * P)
            chans.append(int(im.shape[1]))
        else:
            _req(im, torch.int64, "pack_patches class map"); _req(ce, F32, "pack_patches class_emb")
            want = (B, nh * P, nw * P)
            chans.append(int(ce.shape[1]))
        if tuple(im.shape) != want:
            raise _lib.MmaeError(f"pack_patches: input shape {tuple(im.shape)} != {want}")
    k = ids_keep32.shape[1]
    out = torch.empty((B * k, ktot), dtype=BF16, device=images[0].device)
    T = len(images)
    c_ch = (C.c_int * T)(*chans)
    c_p = (C.c_int * T)(*[int(p) for p in patches])
    c_ncls = (C.c_int * T)(*[0 if ce is None else int(ce.shape[0]) for ce in class_embs])
    _lib.check(_L().mmae_pack_patches(_ptr_array(images), c_ch, c_p, _ptr_array(class_embs), c_ncls, T, B, nh, nw, _ptr(ids_keep32), k,
                                      _ptr(out), ktot, _stream()), "pack_patches")
    return out


def class_emb_bwd(dA: torch.Tensor, class_map: torch.Tensor, ids_keep32: torch.Tensor, tok_start: int, nh: int, nw: int, P: int,
                  num_classes: int, dim_emb: int, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """dA: bf16 [B*k, >= dim_emb*P*P] (a column slice is fine) -> d_emb f32 [num_classes, dim_emb]."""
    _req(class_map, torch.int64, "class_emb_bwd class map")
    if dA.dtype != BF16 or dA.stride(-1) != 1:
        raise _lib.MmaeError("class_emb_bwd: dA must be bf16 with unit column stride")
    B, k = ids_keep32.shape
    if out is None:
        out = torch.empty((num_classes, dim_emb), dtype=F32, device=dA.device)
        accumulate = False
    ws = workspace(_L().mmae_class_emb_bwd_workspace_bytes(num_classes, dim_emb), dA.device)
    _lib.check(_L().mmae_class_emb_bwd(_ptr(dA), dA.stride(0), _ptr(class_map), _ptr(ids_keep32), B, k, tok_start, nh, nw, P,
                                       num_classes, dim_emb, _ptr(out), int(accumulate), _ptr(ws), ws.numel(), _stream()),
               "class_emb_bwd")
    return out


def build_token_table(pos_embs: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], ntok: int, dim: int):
    tab = torch.empty((len(pos_embs) * ntok, dim), dtype=F32, device=pos_embs[0].device)
    _lib.check(_L().mmae_build_token_table(_ptr_array(pos_embs), _ptr_array(biases), len(pos_embs), ntok, dim, _ptr(tab),
                                           _stream()), "build_token_table")
    return tab


def patch_embed_gemm(a: torch.Tensor, wcat: torch.Tensor, add_idx: torch.Tensor, add_tab: torch.Tensor, seg_in: int,
                     seg_out: int, out: torch.Tensor):
    M, K = a.shape
    N = wcat.shape[0]
    with _Timed("gemm_nt", 2.0 * M * N * K, lambda: _nt_label(5)):
        _lib.check(_L().mmae_patch_embed_gemm(_ptr(a), _ptr(wcat), M, N, K, _ptr(add_idx), _ptr(add_tab), seg_in, seg_out,
                                              _ptr(out), out.shape[-1], _stream()), "patch_embed_gemm")
    return out


def append_global(global_tokens: torch.Tensor, x: torch.Tensor, k: int):
    B, seq, dim = x.shape
    if seq == k or B == 0:
        return
    _lib.check(_L().mmae_append_global(_ptr(global_tokens), _ptr(x), B, k, seq - k, dim, _stream()), "append_global")


def embed_bwd_reduce(dx: torch.Tensor, k: int):
    """dx f32 [B, k+G, D] -> (d_global [1,G,D] f32, rows bf16 [B*k, D])."""
    _req(dx, F32, "embed_bwd dx")
    B, seq, dim = dx.shape
    G = seq - k
    d_global = torch.empty((1, max(G, 1), dim), dtype=F32, device=dx.device)
    rows = torch.empty((B * k, dim), dtype=BF16, device=dx.device)
    _lib.check(_L().mmae_embed_bwd_reduce(_ptr(dx), None, B, k, G, dim, 0, 0, _ptr(d_global), None, _ptr(rows), _stream()),
               "embed_bwd_reduce")
    return d_global[:, :G], rows


def build_decoder_emb(task_embs: Sequence[Optional[torch.Tensor]], pos_emb: torch.Tensor, ntok: int, dd: int):
    emb = torch.empty((len(task_embs) * ntok, dd), dtype=F32, device=pos_emb.device)
    _lib.check(_L().mmae_build_decoder_emb(_ptr_array(task_embs), _ptr(pos_emb), len(task_embs), ntok, dd, _ptr(emb), _stream()),
               "build_decoder_emb")
    return emb


def decoder_assemble_fwd(ctx, mask_token, emb, keep32, restore32, k, G, q_start, q_len):
    """ctx: the projected context, bf16 [B, seq, Dd] (f32 for a decoder that runs in fp32)."""
    if ctx.dtype not in (BF16, F32):
        raise _lib.MmaeError(f"decoder_assemble ctx: dtype {ctx.dtype}")
    _req(ctx, ctx.dtype, "decoder_assemble ctx")
    B, seq, dd = ctx.shape
    total = restore32.shape[1]
    queries = torch.empty((B, q_len, dd), dtype=F32, device=ctx.device)
    context = torch.empty((B, seq, dd), dtype=F32, device=ctx.device)
    _lib.check(_L().mmae_decoder_assemble_fwd(_ptr(ctx), int(ctx.dtype == F32), _ptr(mask_token), _ptr(emb), _ptr(keep32), _ptr(restore32),
                                              B, k, G, total, q_start, q_len, dd, _ptr(queries), _ptr(context), _stream()),
               "decoder_assemble_fwd")
    return queries, context


def decoder_assemble_bwd(d_queries, d_context, keep32, restore32, k, G, T, ntok, q_start, q_len, d_ctx_out=None, f32=False):
    """d_ctx_out: optional bf16 [B*seq, Dd] view (unit column stride, any row stride) that receives d_ctx, e.g. a column
    block of a wider matrix; returned reshaped to [B, seq, Dd] only when it is contiguous, else as the 2-D view.
    f32: d_ctx in f32 (a decoder that runs in fp32)."""
    _req(d_queries, F32, "decoder_assemble_bwd d_queries"); _req(d_context, F32, "decoder_assemble_bwd d_context")
    B, seq, dd = d_context.shape
    dev = d_context.device
    if d_ctx_out is None:
        d_ctx, ldc = torch.empty((B, seq, dd), dtype=F32 if f32 else BF16, device=dev), dd
    else:
        _req(d_ctx_out, BF16, "decoder_assemble_bwd d_ctx_out", contiguous=False)
        rows, cols, ldc = _rows2d(d_ctx_out, "decoder_assemble_bwd d_ctx_out")
        if rows != B * seq or cols != dd:
            raise _lib.MmaeError(f"decoder_assemble_bwd: d_ctx_out is {rows}x{cols}, expected {B * seq}x{dd}")
        d_ctx = d_ctx_out
    d_mask = torch.empty((1, 1, dd), dtype=F32, device=dev)
    d_task = [torch.empty((1, 1, dd), dtype=F32, device=dev) for _ in range(T)]
    ws = workspace(_L().mmae_decoder_assemble_bwd_workspace_bytes(B, T, dd), dev)
    _lib.check(_L().mmae_decoder_assemble_bwd(_ptr(d_queries), _ptr(d_context), _ptr(keep32), _ptr(restore32), B, k, G, T, ntok,
                                              q_start, q_len, dd, _ptr(d_ctx), int(d_ctx.dtype == F32), ldc, _ptr(d_mask), _ptr_array(d_task), _ptr(ws),
                                              ws.numel(), _stream()), "decoder_assemble_bwd")
    return d_ctx, d_mask, d_task


def unpatchify_gemm(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, B: int, C_: int, P: int, NH: int, NW: int,
                    out_f32: bool = False):
    M, K, lda = _rows2d(a, "unpatchify a")
    N = w.shape[0]
    img = torch.empty((B, C_, NH * P, NW * P), dtype=F32 if out_f32 else BF16, device=a.device)
    with _Timed("gemm_nt", 2.0 * M * N * K, lambda: _nt_label(6)):
        _lib.check(_L().mmae_unpatchify_gemm(_ptr(a), lda, _ptr(w), M, N, K, _ptr(bias), _ptr(img), int(out_f32), C_, P, NH, NW,
                                             _stream()), "unpatchify_gemm")
    return img


def patchify_bf16(img: torch.Tensor, P: int):
    """image bf16 or f32 [B,C,H,W] -> bf16 rows [B*(H/P)*(W/P), C*P*P]."""
    if img.dtype not in (BF16, F32):
        raise _lib.MmaeError(f"patchify: image dtype {img.dtype}")
    _req(img, img.dtype, "patchify image")
    B, C_, H, W = img.shape
    rows = torch.empty((B * (H // P) * (W // P), C_ * P * P), dtype=BF16, device=img.device)
    _lib.check(_L().mmae_patchify_bf16(_ptr(img), int(img.dtype == F32), B, C_, H // P, W // P, P, _ptr(rows), _stream()), "patchify")
    return rows


# --------------------------------------------------------------------------------------------- losses
def masked_loss(kind: int, norm_pix: bool, pred: torch.Tensor, target: torch.Tensor, mask: Optional[torch.Tensor], patch: int,
                want_grad: bool = True):
    """Returns (loss f32 scalar tensor, d_pred or None)."""
    if pred.dtype not in (BF16, F32):
        raise _lib.MmaeError(f"masked_loss: pred dtype {pred.dtype}")
    _req(pred, pred.dtype, "masked_loss pred"); _req(target, F32, "masked_loss target")
    if mask is not None:
        _req(mask, torch.int64, "masked_loss mask")
    B, C_, H, W = pred.shape
    loss = torch.empty((), dtype=F32, device=pred.device)
    d_pred = torch.empty_like(pred) if want_grad else None
    ws = workspace(_L().mmae_masked_loss_workspace_bytes(B, C_, H, W, patch), pred.device)
    # prediction and target in, gradient out (in the prediction's dtype), the patch mask in
    moved = pred.numel() * (pred.element_size() * (2 if want_grad else 1) + 4) + (mask.numel() * 8 if mask is not None else 0)
    with _Timed("masked_loss", 0.0, "loss_main_kernel (with loss_count_kernel, loss_final_kernel)", nbytes=moved):
        _lib.check(_L().mmae_masked_loss(kind, int(norm_pix), _ptr(pred), int(pred.dtype == BF16), _ptr(target), _ptr(mask), B, C_, H,
                                         W, patch, _ptr(loss), _ptr(d_pred), _ptr(ws), ws.numel(), _stream()), "masked_loss")
    return loss, d_pred


def masked_ce(logits: torch.Tensor, target: torch.Tensor, mask: Optional[torch.Tensor], patch: int, label_smoothing: float = 0.0,
              want_grad: bool = True):
    """Returns (loss f32 scalar tensor, d_logits or None)."""
    if logits.dtype not in (BF16, F32):
        raise _lib.MmaeError(f"masked_ce: logits dtype {logits.dtype}")
    _req(logits, logits.dtype, "masked_ce logits"); _req(target, torch.int64, "masked_ce target")
    B, C_, H, W = logits.shape
    if tuple(target.shape) != (B, H, W):
        raise _lib.MmaeError(f"masked_ce: target shape {tuple(target.shape)} != {(B, H, W)}")
    if mask is not None:
        _req(mask, torch.int64, "masked_ce mask")
        if mask.numel() != B * (H // patch) * (W // patch):
            raise _lib.MmaeError(f"masked_ce: mask has {mask.numel()} entries, expected {B * (H // patch) * (W // patch)}")
    loss = torch.empty((), dtype=F32, device=logits.device)
    d_logits = torch.empty_like(logits) if want_grad else None
    ws = workspace(_L().mmae_masked_ce_workspace_bytes(B, H, W), logits.device)
    moved = logits.numel() * logits.element_size() * (2 if want_grad else 1) + target.numel() * 8
    with _Timed("masked_ce", 0.0, "ce_main_regs_kernel (with its count / final kernels)", nbytes=moved):
        _lib.check(_L().mmae_masked_ce(_ptr(logits), int(logits.dtype == BF16), _ptr(target), _ptr(mask), B, C_, H, W, patch,
                                       float(label_smoothing), _ptr(loss), _ptr(d_logits), _ptr(ws), ws.numel(), _stream()), "masked_ce")
    return loss, d_logits


def standardize_depth(depth: torch.Tensor, lo_frac: float = 0.1, hi_frac: float = 0.9, eps: float = 1e-6,
                      out: Optional[torch.Tensor] = None):
    """Per sample: (x - mean) / sqrt(var + eps) with the moments of the sorted values of ranks [int(lo*n), int(hi*n))."""
    _req(depth, F32, "standardize_depth depth")
    B = depth.shape[0]
    out = torch.empty_like(depth) if out is None else _req(out, F32, "standardize_depth out")
    if B:
        # algorithmic bytes: the map read once and written once (the select's further sweeps of a sample are served by L2)
        with _Timed("standardize_depth", 0.0, "depth_standardize", nbytes=8 * depth.numel()):
            _lib.check(_L().mmae_standardize_depth(_ptr(depth), _ptr(out), B, depth.numel() // B, lo_frac, hi_frac, eps, _stream()),
                       "standardize_depth")
    return out


# --------------------------------------------------------------------------------------------- optimiser
def adamw_flat(p, g, m, v, n_decay, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, w_bf16=None, grad_sumsq=None,
               max_norm=0.0, skip_norm=0.0, step_dev=None, clear_grad=False, sumsq_part=None):
    """clear_grad: g is zeroed behind the read; sumsq_part: f32 [>= adamw_parts(p.numel())] receives partial sums of g^2
    (sum_partials over the parts of every range of a step gives the squared gradient norm)."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _req(t, F32, f"adamw {n}")
    if sumsq_part is not None:
        _req(sumsq_part, F32, "adamw sumsq_part")
        if sumsq_part.numel() < adamw_parts(p.numel()):
            raise ValueError("adamw_flat: sumsq_part too small")
    if grad_sumsq is not None:
        _req(grad_sumsq, F32, "adamw grad_sumsq")
    if step_dev is not None:
        _req(step_dev, torch.int32, "adamw step_dev")
    # p, g, m, v in; p, m, v (and the bf16 weights) out
    with _Timed("adamw", 0.0, "adamw_flat_kernel", nbytes=p.numel() * (28 + (2 if w_bf16 is not None else 0) + (4 if clear_grad else 0))):
        _lib.check(_L().mmae_adamw_flat_ex(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), int(n_decay), lr, beta1, beta2, eps,
                                           weight_decay, int(step), float(grad_scale), _ptr(w_bf16), _ptr(grad_sumsq), float(max_norm or 0.0),
                                           float(skip_norm or 0.0), _ptr(step_dev), int(bool(clear_grad)), _ptr(sumsq_part), _stream()),
                   "adamw_flat")


def adamw_parts(n: int) -> int:
    """Partial sums one adamw_flat call over n elements writes into sumsq_part."""
    return int(_L().mmae_adamw_flat_parts(int(n)))


def sum_partials(part: torch.Tensor, out: Optional[torch.Tensor] = None):
    """Sum of an f32 vector in a fixed order (device scalar)."""
    _req(part, F32, "sum_partials part")
    if out is None:
        out = torch.empty((), dtype=F32, device=part.device)
    _lib.check(_L().mmae_sum_partials(_ptr(part), part.numel(), _ptr(out), _stream()), "sum_partials")
    return out


def sumsq(x: torch.Tensor, out: Optional[torch.Tensor] = None):
    _req(x, F32, "sumsq x")
    if out is None:
        out = torch.empty((), dtype=F32, device=x.device)
    ws = workspace(4096, x.device)
    _lib.check(_L().mmae_sumsq(_ptr(x), x.numel(), _ptr(out), _ptr(ws), ws.numel(), _stream()), "sumsq")
    return out


def cast_transpose(src: torch.Tensor, want: bool = True, want_t: bool = True):
    _req(src, F32, "cast_transpose src")
    rows, cols = src.shape
    d = torch.empty((rows, cols), dtype=BF16, device=src.device) if want else None
    dt = torch.empty((cols, rows), dtype=BF16, device=src.device) if want_t else None
    _lib.check(_L().mmae_cast_transpose(_ptr(src), rows, cols, _ptr(d), _ptr(dt), _stream()), "cast_transpose")
    return d, dt


def cast_bf16(x: torch.Tensor):
    _req(x, F32, "cast_bf16 x")
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    _lib.check(_L().mmae_cast_bf16(_ptr(x), _ptr(y), x.numel(), _stream()), "cast_bf16")
    return y


def transpose_multi(src_flat: torch.Tensor, dst_flat: torch.Tensor, table: torch.Tensor, total_tiles: int):
    """table: int32 [n,4] on the device = (offset, rows, cols, first tile) per 2-D tensor inside the flat bf16 buffers."""
    _req(src_flat, BF16, "transpose_multi src"); _req(dst_flat, BF16, "transpose_multi dst"); _req(table, torch.int32, "transpose_multi table")
    _lib.check(_L().mmae_transpose_multi(_ptr(src_flat), _ptr(dst_flat), _ptr(table), table.shape[0], int(total_tiles), _stream()),
               "transpose_multi")


# --------------------------------------------------------------------------------------------- fp32 path (fp32_output_adapters)
def split_f32(x: torch.Tensor, pattern: int, by_rows: bool = False, block_cols: int = 0):
    """x f32 [rows, cols] (any row stride) -> bf16 blocks hi / lo of x = hi + lo in the order (hi, hi, lo) for pattern 0 and
    (hi, lo, hi) for pattern 1: [rows, 3 * block_cols], or [3 * rows, block_cols] with by_rows; block_cols (default cols) pads
    every block with zero columns."""
    _req(x, F32, "split_f32 x", contiguous=False)
    rows, cols, ld = _rows2d(x, "split_f32 x")
    bc = block_cols or cols
    out = torch.empty((3 * rows, bc) if by_rows else (rows, 3 * bc), dtype=BF16, device=x.device)
    _lib.check(_L().mmae_split_f32(_ptr(x), ld, rows, cols, _ptr(out), int(pattern), int(by_rows), bc, _stream()), "split_f32")
    return out


def gelu_f32(x: torch.Tensor):
    _req(x, F32, "gelu_f32 x")
    y = torch.empty_like(x)
    _lib.check(_L().mmae_gelu_f32(_ptr(x), _ptr(y), x.numel(), _stream()), "gelu_f32")
    return y


def gelu_bwd_f32(x: torch.Tensor, dy: torch.Tensor):
    _req(x, F32, "gelu_bwd_f32 x"); _req(dy, F32, "gelu_bwd_f32 dy")
    dx = torch.empty_like(x)
    _lib.check(_L().mmae_gelu_bwd_f32(_ptr(x), _ptr(dy), _ptr(dx), x.numel(), _stream()), "gelu_bwd_f32")
    return dx


def layernorm_fwd_f32(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-6):
    _req(x, F32, "layernorm_f32 x"); _req(gamma, F32, "layernorm_f32 gamma"); _req(beta, F32, "layernorm_f32 beta")
    dim = x.shape[-1]
    rows = x.numel() // dim
    y = torch.empty_like(x)
    mean = torch.empty((rows,), dtype=F32, device=x.device)
    rstd = torch.empty((rows,), dtype=F32, device=x.device)
    _lib.check(_L().mmae_layernorm_fwd_f32(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), rows, dim, float(eps),
                                           _stream()), "layernorm_fwd_f32")
    return y, mean, rstd


def layernorm_bwd_f32(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor):
    """Returns (dx, dgamma, dbeta), all f32."""
    _req(dy, F32, "layernorm_bwd_f32 dy"); _req(x, F32, "layernorm_bwd_f32 x")
    dim = x.shape[-1]
    rows = x.numel() // dim
    dx = torch.empty_like(x)
    dgamma = torch.empty((dim,), dtype=F32, device=x.device)
    dbeta = torch.empty((dim,), dtype=F32, device=x.device)
    ws = workspace(_L().mmae_layernorm_bwd_workspace_bytes(rows, dim), x.device)
    _lib.check(_L().mmae_layernorm_bwd_f32(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dgamma), _ptr(dbeta), 0,
                                           rows, dim, _ptr(ws), ws.numel(), _stream()), "layernorm_bwd_f32")
    return dx, dgamma, dbeta


def attn_f32_fwd(q, k, v, B, H, q_len, kv_len, head_dim, scale):
    """fp32 attention; q / k / v: f32 row-strided 2-D views as in attn_fwd.  Returns (o f32 [B*q_len, H*d], lse [B,H,q_len])."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _req(t, F32, f"attn_f32_fwd {n}", contiguous=False)
    o = torch.empty((B * q_len, H * head_dim), dtype=F32, device=q.device)
    lse = torch.empty((B, H, q_len), dtype=F32, device=q.device)
    _lib.check(_L().mmae_attn_f32_fwd(_ptr(q), _head_view(q, "q"), _ptr(k), _head_view(k, "k"), _ptr(v), _head_view(v, "v"), _ptr(o),
                                      o.stride(0), _ptr(lse), B, H, q_len, kv_len, head_dim, float(scale), _stream()), "attn_f32_fwd")
    return o, lse


def attn_f32_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, q_len, kv_len, head_dim, scale):
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (d_o, "dO"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _req(t, F32, f"attn_f32_bwd {n}", contiguous=False)
    ws = workspace(_L().mmae_attn_f32_bwd_workspace_bytes(B, H, q_len), q.device)
    _lib.check(_L().mmae_attn_f32_bwd(_ptr(q), _head_view(q, "q"), _ptr(k), _head_view(k, "k"), _ptr(v), _head_view(v, "v"), _ptr(o),
                                      _head_view(o, "o"), _ptr(d_o), _head_view(d_o, "dO"), _ptr(lse), _ptr(dq), _head_view(dq, "dq"),
                                      _ptr(dk), _head_view(dk, "dk"), _ptr(dv), _head_view(dv, "dv"), B, H, q_len, kv_len, head_dim,
                                      float(scale), _ptr(ws), ws.numel(), _stream()), "attn_f32_bwd")
