"""Autograd functions that chain the HIP kernels into the pieces of the MultiMAE step.

Precision plan (the reference's AMP semantics with bf16 as the low-precision type): parameters, the residual
stream, LayerNorm statistics, softmax and the losses are fp32; every matrix product takes bf16 operands and
accumulates in fp32.  Each function saves exactly what its backward kernels read."""
from typing import Dict, List, Optional, Sequence, Tuple

import os

import torch

from . import ops

# ---------------------------------------------------------------------------------------------
# bf16 copies of the fp32 master weights ([out,in] for the forward / dW, [in,out] for dA)
# ---------------------------------------------------------------------------------------------
_WEIGHT_EPOCH = 0
_WEIGHT_CACHE: Dict[int, tuple] = {}


def bump_weight_epoch():
    """Called by optimisers that update parameters through raw pointers (no autograd version bump)."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


def bf16_weights(w: torch.Tensor):
    """(w as bf16 [N,K], w^T as bf16 [K,N]) for a 2-D fp32 parameter, cached until the parameter changes.
    Parameters in flat storage carry views of the flat bf16 copies that the optimizer step refreshes in two launches."""
    views = getattr(w, "_mmae_bf16_views", None)
    if views is not None:
        if w._version != views[2]:  # written through autograd-visible ops (load_state_dict, copy_) since the last sync
            views[3]()
            views = w._mmae_bf16_views
        return views[0], views[1]
    key = id(w)
    ent = _WEIGHT_CACHE.get(key)
    ver = (w._version, _WEIGHT_EPOCH, w.data_ptr())
    if ent is not None and ent[0] == ver and ent[1] is w:
        return ent[2], ent[3]
    wb, wt = ops.cast_transpose(w.detach().reshape(w.shape[0], -1))
    _WEIGHT_CACHE[key] = (ver, w, wb, wt)
    return wb, wt


def clear_weight_cache():
    _WEIGHT_CACHE.clear()


# bf16 shadow of the fp32 gradient that a LayerNorm backward just produced: the next consumer needs both precisions
# (fp32 for the residual path, bf16 as a GEMM operand).  The shadow travels with the fp32 tensor itself, as an attribute
# (a tensor keeps its Python object for as long as it lives, so the attribute survives the trip through the autograd
# engine); a consumer that holds a whole-tensor view of it finds it through the view's base.  A gradient that the engine
# summed with another, or that anything modified in place since, carries no (valid) shadow and is cast.
def _set_shadow(dx: torch.Tensor, dxb: torch.Tensor):
    dx._mmae_shadow = (dxb, dx._version)


def _bf16(g: torch.Tensor) -> torch.Tensor:
    if g.dtype == torch.bfloat16:
        return g
    root = g._base if g._base is not None else g
    ent = getattr(root, "_mmae_shadow", None)
    if ent is not None:
        root._mmae_shadow = None  # one consumer; the bf16 copy is not kept alive beyond it
        if (ent[1] == g._version and g.is_contiguous() and g.numel() == root.numel()
                and g.storage_offset() == root.storage_offset()):  # all of it, unmodified since the shadow was made
            return ent[0].reshape(g.shape)
    return ops.cast_bf16(g.contiguous())


def _direct(p) -> bool:
    """True when the parameter's gradient lives in flat storage that the kernels may accumulate into directly
    (set up by multimae_amd.train.FlatParams); autograd's own accumulation step is then skipped for it."""
    return p is not None and getattr(p, "_mmae_direct", False) and p.grad is not None


def _ready(p):
    cb = getattr(p, "_mmae_ready", None) if p is not None else None
    if cb is not None:
        cb()


def _deliver(p, g):
    """Hand gradient `g` of parameter `p` over: added into the flat gradient storage here, on the stream the backward
    kernel ran on, when the parameter lives there (the Function then returns None for it and autograd's own accumulation
    node, which is bound to the stream of the iteration that created it, never sees a tensor); returned otherwise."""
    if g is None or not _direct(p):
        return g
    p.grad.add_(g.reshape(p.grad.shape))
    _ready(p)
    return None


# Weight gradients that accumulate straight into flat storage do not feed the rest of backward, so they run on a second
# stream: their workgroups fill the CUs that the main chain's kernels leave idle in their last, partly filled round.
SIDE_STREAM_DW = os.environ.get("MMAE_SIDE_STREAM", "1") != "0"
_SIDE = {}  # device index -> [stream, work pending since the last join]


_TAKEN = {}  # device index -> streams already handed out by concurrent_stream()


def concurrent_stream(device) -> torch.cuda.Stream:
    """A new stream whose kernels run beside those of the current stream and of the streams handed out before.
    A process has 4 hardware queues; which queue a stream lands on depends on how many streams the process (PyTorch,
    RCCL) created before it, and two streams on one queue do not overlap at all: a weight-gradient stream that shares
    the main stream's queue costs 4 ms of a 29 ms step.  Candidates are tried with pairs of idle timed kernels: side by
    side they take the time of one, on one queue the time of two.  Runs once per stream, at first use (it
    synchronises the device)."""
    from . import ops, _lib
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    others = [torch.cuda.current_stream(idx)] + _TAKEN.setdefault(idx, [])
    lib, us = _lib.load(), 300
    best = None
    for _ in range(8):
        cand = torch.cuda.Stream(device=idx)
        torch.cuda.synchronize(idx)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        main = others[0]
        e0.record(main)
        for st in others + [cand]:
            st.wait_stream(main)  # start together
        for st in others + [cand]:
            _lib.check(lib.mmae_spin_us(us, st.cuda_stream), "spin_us")
        for st in others[1:] + [cand]:
            main.wait_stream(st)
        e1.record(main)
        torch.cuda.synchronize(idx)
        ms = e0.elapsed_time(e1)
        if best is None or ms < best[0]:
            best = (ms, cand)
        if ms < 1.5 * us * 1e-3:  # all of them side by side
            break
    _TAKEN[idx].append(best[1])
    return best[1]


def prepare_streams(device, n_decoders: int = 0):
    """Create the weight-gradient stream and the decoder stream of `device` now, from the stream the step runs on (they
    are chosen against the current stream, and backward nodes run under other streams)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if SIDE_STREAM_DW and dev.index not in _SIDE:
        _side(dev)
    if n_decoders > 1 and dev.index not in _AUX:
        if decoder_lanes(dev, n_decoders) is not None:
            _AUX[dev.index][1] = False  # nothing queued on it yet


def _side(device):
    ent = _SIDE.get(device.index)
    if ent is None:
        ent = _SIDE[device.index] = [concurrent_stream(device), False]
    return ent


def join_side_stream(device=None):
    """Make the current stream wait for the weight-gradient stream and the decoder streams (call before gradients are
    read: after backward, before the gradient norm, the optimizer step or zero_grad)."""
    join_aux_streams(device)
    for idx, ent in _SIDE.items():
        if ent[1] and (device is None or torch.device(device).index in (None, idx)):
            torch.cuda.current_stream(idx).wait_stream(ent[0])
            ent[1] = False


# The output adapters (decoders) are independent of each other between the encoder output and the losses, and their
# GEMMs are small (196 tiles of 256x256 on 256 CUs, short K): they are dealt over two streams so that the workgroups of
# one decoder fill the CUs another leaves idle.  Autograd runs every backward node on the stream its forward ran on.
DECODER_STREAMS = os.environ.get("MMAE_DEC_STREAMS", "1") != "0"
_DEC_LAYOUT = os.environ.get("MMAE_DEC_LAYOUT", "")  # "<aux streams>[+main]" overrides the default below (tuning)
_AUX = {}  # device index -> [decoder streams, used since the last join, the stream that shares the decoders with them]


def decoder_lanes(device, n_decoders: int):
    """Streams the decoders are dealt over, or None for "all on the current stream": the current stream and one decoder
    stream, decoders alternating.  A process gets 4 hardware queues (GPU_MAX_HW_QUEUES; more are time-sliced and cost a
    third of the throughput); streams beyond that share a queue, and kernels of streams that share a queue run in
    submission order.  With the weight-gradient stream and, in a multi-rank run, the RCCL stream that makes four:
    no compute stream ever queues behind an all-reduce kernel waiting for its peers.  (One stream per decoder measured
    the same as this layout, 29.7 ms per step on one box.)"""
    if not DECODER_STREAMS or n_decoders < 2:
        return None
    if _DEC_LAYOUT:
        n_aux, with_main = int(_DEC_LAYOUT.split("+")[0]), _DEC_LAYOUT.endswith("+main")
    else:
        n_aux, with_main = 1, True
    aux = aux_streams(device, min(n_aux, n_decoders))
    if not with_main:
        return aux
    main = torch.cuda.current_stream(device)
    _AUX[main.device.index][2] = main  # a join issued from a decoder stream has to wait for this one as well
    return [main] + aux


def aux_streams(device, n: int):
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    ent = _AUX.get(idx)
    if ent is None:
        ent = _AUX[idx] = [[], False, None]
    while len(ent[0]) < n:
        ent[0].append(concurrent_stream(torch.device("cuda", idx)))
    ent[1] = True
    return ent[0][:n]


def join_aux_streams(device=None, reset: bool = True):
    """Make the current stream wait for every stream the decoders ran on (before anything reads what their kernels wrote
    in place).  The current stream may itself be a decoder stream: a gradient bucket can complete inside a decoder's
    backward, while kernels of another decoder that write the same bucket are still queued on the main stream."""
    for idx, ent in _AUX.items():
        if ent[1] and (device is None or torch.device(device).index in (None, idx)):
            cur = torch.cuda.current_stream(idx)
            for st in ent[0] + ([ent[2]] if ent[2] is not None else []):
                if st != cur:
                    cur.wait_stream(st)
            if reset:
                ent[1] = False


def side_stream_pending(device) -> Optional[torch.cuda.Stream]:
    ent = _SIDE.get(torch.device(device).index)
    return ent[0] if ent is not None and ent[1] else None


def _lin_grads(dy_b: torch.Tensor, x_b: torch.Tensor, w=None, b=None):
    """(dW [N,K] f32, db [N] f32) of y = x W^T + b from bf16 dy [M,N] and x [M,K]; (None, None) when they were
    accumulated straight into w.grad / b.grad."""
    want_b = b is not None and b.requires_grad
    if _direct(w) and (not want_b or _direct(b)):
        if SIDE_STREAM_DW:
            ent = _side(dy_b.device)
            side = ent[0]
            side.wait_stream(torch.cuda.current_stream(dy_b.device))
            with torch.cuda.stream(side):
                ops.gemm_tn(dy_b, x_b, out=w.grad, accumulate=True, colsum_out=b.grad if want_b else None, colsum_accumulate=True)
            dy_b.record_stream(side)
            x_b.record_stream(side)
            ent[1] = True
        else:
            ops.gemm_tn(dy_b, x_b, out=w.grad, accumulate=True, colsum_out=b.grad if want_b else None, colsum_accumulate=True)
        _ready(w)
        if want_b:
            _ready(b)
        return None, None
    if want_b:
        return ops.gemm_tn(dy_b, x_b, colsum_out=True)
    return ops.gemm_tn(dy_b, x_b), None


def _ln_bwd(dy, x, n_w, n_b, mean, rstd, dres=None, shadow=False):
    """LayerNorm backward; (dx, dgamma, dbeta) with the parameter gradients None when written in place."""
    if _direct(n_w) and _direct(n_b):
        if SIDE_STREAM_DW:
            # the parameter-gradient reduction (a few dozen workgroups) leaves the main chain: it would queue behind the
            # weight-gradient workgroups that own the CUs
            dx, dxb, part, _ = ops.layernorm_bwd(dy, x, n_w, mean, rstd, dres=dres, want_bf16=shadow, defer_params=True)
            ent = _side(x.device)
            ent[0].wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(ent[0]):
                ops.layernorm_param_reduce(part, x.numel() // x.shape[-1], x.shape[-1], n_w.grad, n_b.grad, True)
            part.record_stream(ent[0])
            ent[1] = True
        else:
            dx, dxb, _, _ = ops.layernorm_bwd(dy, x, n_w, mean, rstd, dres=dres, want_bf16=shadow, dgamma=n_w.grad, dbeta=n_b.grad)
        _ready(n_w)
        _ready(n_b)
        dg = db = None
    else:
        dx, dxb, dg, db = ops.layernorm_bwd(dy, x, n_w, mean, rstd, dres=dres, want_bf16=shadow)
    if shadow:
        _set_shadow(dx, dxb)
    return dx, dg, db


# ---------------------------------------------------------------------------------------------
# pieces shared by the encoder block, the decoder blocks and the decoder head
# ---------------------------------------------------------------------------------------------
def _mlp_fwd(x, n_w, n_b, fc1_w, fc1_b, fc2_w, fc2_b, eps=1e-6, pre_ln=None, next_norm=None):
    """x f32 [M,D] -> x + fc2(gelu(fc1(LN(x)))) and the tensors the backward needs.  pre_ln = (LN(x) bf16, mean, rstd) when
    the product that made x already normalised it (ops.gemm_nt_resid_ln)."""
    ln, mean, rstd = pre_ln if pre_ln is not None else ops.layernorm_fwd(x, n_w, n_b, eps)
    w1, w1t = bf16_weights(fc1_w)
    w2, w2t = bf16_weights(fc2_w)
    h, act = ops.gemm_nt(ops.EPI_GELU, ln, w1, fc1_b)
    saved = (x, ln, mean, rstd, h, act, w1t, w2t, (n_w, n_b, fc1_w, fc1_b, fc2_w, fc2_b))
    if next_norm is not None:  # callers that pass it take three results
        D = x.shape[-1]
        if D <= 1024 and D % 256 == 0:
            y, ln_n, mean_n, rstd_n = ops.gemm_nt_resid_ln(act, w2, fc2_b, x, next_norm[0], next_norm[1], next_norm[2])
            return y, saved, (ln_n, mean_n, rstd_n)
        return ops.gemm_nt(ops.EPI_RESID, act, w2, fc2_b, aux=x), saved, None
    y = ops.gemm_nt(ops.EPI_RESID, act, w2, fc2_b, aux=x)
    return y, saved


def _mlp_bwd(g, saved):
    """g f32 [M,D] (gradient of the block output) -> (dx f32, [dn_w, dn_b, dfc1_w, dfc1_b, dfc2_w, dfc2_b])."""
    x, ln, mean, rstd, h, act, w1t, w2t, (n_w, n_b, fc1_w, fc1_b, fc2_w, fc2_b) = saved
    gb = _bf16(g)
    dh = ops.gemm_nt(ops.EPI_DGELU, gb, w2t, None, aux=h)
    dw2, db2 = _lin_grads(gb, act, fc2_w, fc2_b)
    dln = ops.gemm_nt(ops.EPI_BF16, dh, w1t, None)
    dw1, db1 = _lin_grads(dh, ln, fc1_w, fc1_b)
    dx, dnw, dnb = _ln_bwd(dln, x, n_w, n_b, mean, rstd, dres=g, shadow=True)  # consumed right after (attention half / decoder head)
    return dx, [dnw, dnb, dw1, db1, dw2, db2]


def _self_attn_fwd(x, B, N, heads, n_w, n_b, qkv_w, qkv_b, proj_w, proj_b, eps=1e-6, next_norm=None, pre_ln=None):
    """next_norm = (gamma, beta, eps) of the LayerNorm that reads this half's output (the block's norm2): the residual product
    then also returns (LN(y) bf16, mean, rstd) as a third result (None without it)."""
    D = x.shape[-1]
    d = D // heads
    ln, mean, rstd = pre_ln if pre_ln is not None else ops.layernorm_fwd(x, n_w, n_b, eps)
    wq, wqt = bf16_weights(qkv_w)
    wp, wpt = bf16_weights(proj_w)
    qkv = ops.gemm_nt(ops.EPI_BF16, ln, wq, qkv_b)
    o, lse = ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, heads, N, N, d, d ** -0.5)
    saved = (x, ln, mean, rstd, qkv, o, lse, wqt, wpt, (n_w, n_b, qkv_w, qkv_b, proj_w, proj_b), B, N, heads)
    if next_norm is not None and D <= 1024 and D % 256 == 0:
        y, ln2, mean2, rstd2 = ops.gemm_nt_resid_ln(o, wp, proj_b, x, next_norm[0], next_norm[1], next_norm[2])
        return y, saved, (ln2, mean2, rstd2)
    y = ops.gemm_nt(ops.EPI_RESID, o, wp, proj_b, aux=x)
    return y, saved, None


def _self_attn_bwd(g, saved):
    x, ln, mean, rstd, qkv, o, lse, wqt, wpt, (n_w, n_b, qkv_w, qkv_b, proj_w, proj_b), B, N, heads = saved
    D = x.shape[-1]
    d = D // heads
    gb = _bf16(g)
    d_o = ops.gemm_nt(ops.EPI_BF16, gb, wpt, None)
    dwp, dbp = _lin_grads(gb, o, proj_w, proj_b)
    dqkv = torch.empty_like(qkv)
    ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, d_o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                 B, heads, N, N, d, d ** -0.5)
    dln = ops.gemm_nt(ops.EPI_BF16, dqkv, wqt, None)
    dwq, dbq = _lin_grads(dqkv, ln, qkv_w, qkv_b)
    dx, dnw, dnb = _ln_bwd(dln, x, n_w, n_b, mean, rstd, dres=g, shadow=True)  # consumed by the previous block
    return dx, [dnw, dnb, dwq, dbq, dwp, dbp]


class BlockFn(torch.autograd.Function):
    """Pre-LN transformer block (multimae/multimae_utils.py:282-285 of the reference), x f32 [B,N,D]; eps1 / eps2 are the
    eps of the two LayerNorms."""

    @staticmethod
    def forward(ctx, x, heads, n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b, eps1=1e-6, eps2=1e-6, chain=None):
        """chain (run_blocks): a dict that carries the first LayerNorm of a block from the block before it, whose last
        product computes it: chain["pre"] = (norm1(x) bf16, mean, rstd) of THIS block's input when present (consumed here);
        chain["next"] = (gamma, beta, eps) of the next block's norm1, for which this block leaves chain["pre"]."""
        B, N, D = x.shape
        x2 = x.contiguous().reshape(B * N, D)
        pre = chain.pop("pre", None) if chain is not None else None
        nxt = chain.get("next") if chain is not None else None
        x1, s_att, ln2 = _self_attn_fwd(x2, B, N, heads, n1w, n1b, qkvw, qkvb, pw, pb, eps1, next_norm=(n2w, n2b, eps2), pre_ln=pre)
        if nxt is not None:
            y, s_mlp, ln_next = _mlp_fwd(x1, n2w, n2b, f1w, f1b, f2w, f2b, eps2, pre_ln=ln2, next_norm=nxt)
        else:
            (y, s_mlp), ln_next = _mlp_fwd(x1, n2w, n2b, f1w, f1b, f2w, f2b, eps2, pre_ln=ln2), None
        if ln_next is not None:
            chain["pre"] = ln_next
        ctx.saved = (s_att, s_mlp)
        ctx.shape = (B, N, D)
        return y.reshape(B, N, D)

    @staticmethod
    def backward(ctx, g):
        B, N, D = ctx.shape
        s_att, s_mlp = ctx.saved
        ctx.saved = None
        g1, gm = _mlp_bwd(g.contiguous().reshape(B * N, D), s_mlp)
        g0, ga = _self_attn_bwd(g1, s_att)
        return (g0.reshape(B, N, D), None, *ga, *gm, None, None, None)


def run_blocks(blocks, x):
    """x through a sequence of multimae_utils.Block modules (the encoder).  Blocks that run as the fused node hand the first
    LayerNorm of the next block to their own last product (ops.gemm_nt_resid_ln: idle CUs of that product's tail launch
    normalise the rows that are complete); results are bitwise those of calling the blocks one after the other."""
    blocks = list(blocks)
    chain = {}
    for i, blk in enumerate(blocks):
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        fused = getattr(blk, "fused_node", None)
        if fused is None or not fused():
            chain.clear()
            x = blk(x)
            continue
        chain.pop("next", None)
        nfused = getattr(nxt, "fused_node", None) if nxt is not None else None
        if nfused is not None and nfused() and x.is_cuda:
            chain["next"] = (nxt.norm1.weight.detach(), nxt.norm1.bias.detach(), nxt.norm1.eps)
        x = blk(x, chain=chain)
    return x


# ---------------------------------------------------------------------------------------------
# the same pieces as stand-alone modules (no LayerNorm / residual around them): Mlp, Attention, CrossAttention
# called directly, as the reference's classes can be (multimae/multimae_utils.py:162-169, :191-207, :224-247)
# ---------------------------------------------------------------------------------------------
def _rows_bf16(x: torch.Tensor) -> torch.Tensor:
    """x [..., D] as contiguous bf16 rows [M, D]: cast from f32, taken as it is when already bf16 (LayerNormFn's output)."""
    x2 = x.contiguous().reshape(-1, x.shape[-1])
    return x2 if x2.dtype == torch.bfloat16 else ops.cast_bf16(x2)


def _grad_epi(x: torch.Tensor) -> int:
    """Epilogue of the product that forms the gradient of input x: the gradient has the input's dtype."""
    return ops.EPI_BF16 if x.dtype == torch.bfloat16 else ops.EPI_F32


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension as a node of its own: x f32 [..., D] -> bf16 [..., D] (the operand type of the
    product that follows it; MlpFn / AttentionFn / CrossAttentionFn take it without a cast and hand a bf16 gradient back).
    The fused blocks carry their LayerNorms inside; this node serves the unfused paths (drop rates > 0 in training mode,
    decoders without cross-attention or with their own queries)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x2 = x.contiguous().reshape(-1, x.shape[-1])
        y, mean, rstd = ops.layernorm_fwd(x2, w, b, eps)
        ctx.saved = (x2, mean, rstd, (w, b), x.shape)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, mean, rstd, (w, b), shape = ctx.saved
        ctx.saved = None
        dx, dw, db = _ln_bwd(_bf16(g.contiguous().reshape(x2.shape)), x2, w, b, mean, rstd)
        return dx.reshape(shape), dw, db, None


class LinearFn(torch.autograd.Function):
    """x W^T + b as a node of its own, x f32 or bf16 [..., K] -> f32 [..., N]: serves an Mlp whose activation is not the
    exact GELU that the fc1 epilogue computes (the activation then runs as its torch module between two of these)."""

    @staticmethod
    def forward(ctx, x, w, b):
        xb = _rows_bf16(x)
        wb, wt = bf16_weights(w)
        y = ops.gemm_nt(ops.EPI_F32, xb, wb, b)
        ctx.saved = (xb, wt, (w, b), x.shape)
        ctx.in_epi = _grad_epi(x)
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, g):
        xb, wt, (w, b), shape = ctx.saved
        ctx.saved = None
        gb = _bf16(g.contiguous().reshape(-1, g.shape[-1]))
        dx = ops.gemm_nt(ctx.in_epi, gb, wt, None)
        dw, db = _lin_grads(gb, xb, w, b)
        return dx.reshape(shape), dw, db


class MlpFn(torch.autograd.Function):
    """fc2(gelu(fc1(x))), x f32 [..., D] -> f32 [..., D_out]."""

    @staticmethod
    def forward(ctx, x, fc1_w, fc1_b, fc2_w, fc2_b):
        xb = _rows_bf16(x)
        w1, w1t = bf16_weights(fc1_w)
        w2, w2t = bf16_weights(fc2_w)
        h, act = ops.gemm_nt(ops.EPI_GELU, xb, w1, fc1_b)
        y = ops.gemm_nt(ops.EPI_F32, act, w2, fc2_b)
        ctx.saved = (xb, h, act, w1t, w2t, (fc1_w, fc1_b, fc2_w, fc2_b), x.shape)
        ctx.in_epi = _grad_epi(x)
        return y.reshape(*x.shape[:-1], fc2_w.shape[0])

    @staticmethod
    def backward(ctx, g):
        xb, h, act, w1t, w2t, (fc1_w, fc1_b, fc2_w, fc2_b), shape = ctx.saved
        ctx.saved = None
        gb = _bf16(g.contiguous().reshape(-1, g.shape[-1]))
        dh = ops.gemm_nt(ops.EPI_DGELU, gb, w2t, None, aux=h)
        dw2, db2 = _lin_grads(gb, act, fc2_w, fc2_b)
        dx = ops.gemm_nt(ctx.in_epi, dh, w1t, None)
        dw1, db1 = _lin_grads(dh, xb, fc1_w, fc1_b)
        return dx.reshape(shape), dw1, db1, dw2, db2


class AttentionFn(torch.autograd.Function):
    """proj(softmax(q k^T scale) v) with q, k, v = split(qkv(x)), x f32 [B,N,D]."""

    @staticmethod
    def forward(ctx, x, heads, qkv_w, qkv_b, proj_w, proj_b):
        B, N, D = x.shape
        d = D // heads
        xb = _rows_bf16(x)
        wq, wqt = bf16_weights(qkv_w)
        wp, wpt = bf16_weights(proj_w)
        qkv = ops.gemm_nt(ops.EPI_BF16, xb, wq, qkv_b)
        o, lse = ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, heads, N, N, d, d ** -0.5)
        y = ops.gemm_nt(ops.EPI_F32, o, wp, proj_b)
        ctx.saved = (xb, qkv, o, lse, wqt, wpt, (qkv_w, qkv_b, proj_w, proj_b), (B, N, D, heads))
        ctx.in_epi = _grad_epi(x)
        return y.reshape(B, N, D)

    @staticmethod
    def backward(ctx, g):
        xb, qkv, o, lse, wqt, wpt, (qkv_w, qkv_b, proj_w, proj_b), (B, N, D, heads) = ctx.saved
        ctx.saved = None
        d = D // heads
        gb = _bf16(g.contiguous().reshape(B * N, D))
        d_o = ops.gemm_nt(ops.EPI_BF16, gb, wpt, None)
        dwp, dbp = _lin_grads(gb, o, proj_w, proj_b)
        dqkv = torch.empty_like(qkv)
        ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, d_o, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                     B, heads, N, N, d, d ** -0.5)
        dx = ops.gemm_nt(ctx.in_epi, dqkv, wqt, None)
        dwq, dbq = _lin_grads(dqkv, xb, qkv_w, qkv_b)
        return dx.reshape(B, N, D), None, dwq, dbq, dwp, dbp


class CrossAttentionFn(torch.autograd.Function):
    """proj(softmax(q(x) k(ctx)^T scale) v(ctx)), x f32 [B,N,D], context f32 [B,M,D]."""

    @staticmethod
    def forward(ctx, x, context, heads, q_w, q_b, kv_w, kv_b, proj_w, proj_b):
        B, N, D = x.shape
        M = context.shape[1]
        d = D // heads
        xb = _rows_bf16(x)
        cb = _rows_bf16(context)
        ctx.in_epi = (_grad_epi(x), _grad_epi(context))
        wq, wqt = bf16_weights(q_w)
        wkv, wkvt = bf16_weights(kv_w)
        wp, wpt = bf16_weights(proj_w)
        q = ops.gemm_nt(ops.EPI_BF16, xb, wq, q_b)
        kv = ops.gemm_nt(ops.EPI_BF16, cb, wkv, kv_b)
        o, lse = ops.attn_fwd(q, kv[:, :D], kv[:, D:], B, heads, N, M, d, d ** -0.5)
        y = ops.gemm_nt(ops.EPI_F32, o, wp, proj_b)
        ctx.saved = (xb, cb, q, kv, o, lse, wqt, wkvt, wpt, (q_w, q_b, kv_w, kv_b, proj_w, proj_b), (B, N, M, D, heads))
        return y.reshape(B, N, D)

    @staticmethod
    def backward(ctx, g):
        xb, cb, q, kv, o, lse, wqt, wkvt, wpt, (q_w, q_b, kv_w, kv_b, proj_w, proj_b), (B, N, M, D, heads) = ctx.saved
        ctx.saved = None
        d = D // heads
        gb = _bf16(g.contiguous().reshape(B * N, D))
        d_o = ops.gemm_nt(ops.EPI_BF16, gb, wpt, None)
        dwp, dbp = _lin_grads(gb, o, proj_w, proj_b)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        ops.attn_bwd(q, kv[:, :D], kv[:, D:], o, d_o, lse, dq, dkv[:, :D], dkv[:, D:], B, heads, N, M, d, d ** -0.5)
        dx = ops.gemm_nt(ctx.in_epi[0], dq, wqt, None)
        dctx = ops.gemm_nt(ctx.in_epi[1], dkv, wkvt, None)
        dwq, dbq = _lin_grads(dq, xb, q_w, q_b)
        dwkv, dbkv = _lin_grads(dkv, cb, kv_w, kv_b)
        return dx.reshape(B, N, D), dctx.reshape(B, M, D), None, dwq, dbq, dwkv, dbkv, dwp, dbp


# ---------------------------------------------------------------------------------------------
# input side: masks, patch projection of the kept tokens, global tokens
# ---------------------------------------------------------------------------------------------
def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _token_table_grad(dx: torch.Tensor, keep32: torch.Tensor, k: int, total: int) -> torch.Tensor:
    """Gradient of the per-token additive table [total, D] (position embeddings) from dx f32 [B, k+G, D]: entry p sums, over
    the samples that kept token p, the row of dx at p's rank.  Gather form in a fixed order (no atomics), 32 samples at a
    time: only models with learnable position embeddings come here."""
    B, _, D = dx.shape
    out = torch.zeros((total, D), dtype=torch.float32, device=dx.device)
    ar = torch.arange(k, device=dx.device).unsqueeze(0)
    for b0 in range(0, B, 32):
        kb = keep32[b0:b0 + 32].long()
        inv = torch.full((kb.shape[0], total), k, dtype=torch.int64, device=dx.device)
        inv.scatter_(1, kb, ar.expand(kb.shape[0], k))
        rows = torch.cat([dx[b0:b0 + 32, :k], dx.new_zeros(kb.shape[0], 1, D)], dim=1)
        out += torch.gather(rows, 1, inv.unsqueeze(-1).expand(-1, -1, D)).sum(dim=0)
    return out


class EmbedFn(torch.autograd.Function):
    """Kept-token patch embedding: Conv2d(k=s=P) + pos-emb + gather(ids_keep) + global tokens
    (multimae/input_adapters.py:126-135, multimae/multimae.py:393-406 of the reference) -> x f32 [B, k+G, D].
    Only the kept tokens are projected (the reference projects all of them and drops the rest).
    A class-map domain (int64 input + class-embedding table; the SemSegInputAdapter named at
    run_pretraining_multimae.py:72-85) is embedded while its patches are packed.

    apply(keep32, patches, grid, pos_tables, global_tokens, n, *inputs[n], *proj_weights[n], *proj_biases[n], *class_embs[n],
          *learnable_pos[n])
    with class_embs[t] = None for image domains and learnable_pos[t] = pos_tables[t] again where the position table takes a
    gradient (learnable_pos_emb, or sincos_pos_emb=False; multimae/input_adapters.py:84-95), None elsewhere (the group
    may be left out altogether)."""

    @staticmethod
    def forward(ctx, keep32, patches, grid, pos_tables, global_tokens, n_img, *tensors):
        ctx.n_tensors = len(tensors)
        images = tensors[:n_img]
        weights = tensors[n_img:2 * n_img]
        biases = tensors[2 * n_img:3 * n_img]
        class_embs = tensors[3 * n_img:4 * n_img]
        pos_learn = tensors[4 * n_img:5 * n_img] or (None,) * n_img
        nh, nw = grid
        B = images[0].shape[0]
        D = weights[0].shape[0]
        k = keep32.shape[1]
        G = global_tokens.shape[1]
        ntok = nh * nw
        widths = [w[0].numel() for w in weights]
        kpatch = sum(widths)
        ktot = round_up(kpatch + 8, 64)
        a = ops.pack_patches([im.contiguous() for im in images], patches, keep32, ktot, nh, nw,
                             [None if ce is None else ce.detach().contiguous() for ce in class_embs])
        wcat = torch.zeros((D, ktot), dtype=torch.bfloat16, device=a.device)
        off = 0
        for w, n in zip(weights, widths):
            wcat[:, off:off + n] = w.detach().reshape(D, n)
            off += n
        tab = ops.build_token_table([p.detach().contiguous() for p in pos_tables], [b_.detach() for b_ in biases], ntok, D)
        x = torch.empty((B, k + G, D), dtype=torch.float32, device=a.device)
        if k > 0:
            ops.patch_embed_gemm(a, wcat, keep32, tab, k, k + G, x)
        ops.append_global(global_tokens.detach().contiguous(), x, k)
        class_maps = [im if ce is not None else None for im, ce in zip(images, class_embs)]
        ctx.saved = (a, widths, [w.shape for w in weights], k, n_img, keep32, class_maps, weights, patches, (nh, nw),
                     [None if ce is None else tuple(ce.shape) for ce in class_embs], (global_tokens, biases, class_embs), pos_learn)
        return x

    @staticmethod
    def backward(ctx, dx):
        (a, widths, wshapes, k, n_img, keep32, class_maps, weights, patches, (nh, nw), ce_shapes, (global_tokens, biases, class_embs),
         pos_learn) = ctx.saved
        ctx.saved = None
        dx = dx.contiguous()
        d_pos = [None] * n_img
        if any(p is not None for p in pos_learn):
            table = _token_table_grad(dx, keep32, k, n_img * nh * nw)  # [n_img * ntok, D]
            for t, p in enumerate(pos_learn):
                if p is not None:
                    d_pos[t] = _deliver(p, table[t * nh * nw:(t + 1) * nh * nw].t().reshape(1, -1, nh, nw))
        d_global, rows = ops.embed_bwd_reduce(dx, k)
        dwcat = ops.gemm_tn(rows, a)  # [D, ktot]: patch columns, then the one-hot(domain) columns = bias grads
        kpatch = sum(widths)
        dws, dbs, dces, off = [], [], [], 0
        for t, (n, shp) in enumerate(zip(widths, wshapes)):
            dws.append(dwcat[:, off:off + n].reshape(shp))
            dbs.append(dwcat[:, kpatch + t].contiguous())
            if class_maps[t] is None:
                dces.append(None)
            else:
                # gradient of this domain's packed rows, then a fixed-order scatter to the embedding rows by class id
                _, wt = bf16_weights(weights[t])
                d_rows = ops.gemm_nt(ops.EPI_BF16, rows, wt, None)
                ncls, cdim = ce_shapes[t]
                dces.append(ops.class_emb_bwd(d_rows, class_maps[t].contiguous(), keep32, t * nh * nw, nh, nw, patches[t], ncls, cdim))
            off += n
        d_global = _deliver(global_tokens, d_global)
        dws = [_deliver(w, d) for w, d in zip(weights, dws)]
        dbs = [_deliver(b_, d) for b_, d in zip(biases, dbs)]
        dces = [_deliver(ce, d) for ce, d in zip(class_embs, dces)]
        return (None, None, None, None, d_global, None, *([None] * n_img), *dws, *dbs, *dces, *d_pos)[:6 + ctx.n_tensors]


# ---------------------------------------------------------------------------------------------
# decoder head: proj_context + query/context assembly + cross-attention + MLP
# ---------------------------------------------------------------------------------------------
class ContextFan:
    """What the decoders of one forward pass share for the gradient of the encoder output.  Every decoder projects the same
    encoder tokens (proj_context, multimae/output_adapters.py:341 of the reference), so d(enc) = sum_i d_ctx_i W_i: instead
    of one product and one fp32 [B, seq, D] result per decoder plus the additions between them, the decoders write their d_ctx
    side by side as the column blocks of one bf16 matrix [B * seq, n * Dd] and ContextFanFn.backward forms the sum as ONE
    product with the row-concatenated weights (K = n * Dd)."""

    def __init__(self, n: int):
        self.n, self.dctx, self.wcat, self.written = n, None, None, [False] * n

    def slot(self, i: int, rows: int, dd: int, wt: torch.Tensor):
        """Decoder i's column block of the d_ctx matrix; its weight block W_i^T [D, Dd] is copied into the K-concatenated
        weight now, on the decoder's stream: once the decoder has signalled its weight gradient, an optimizer that runs beside
        backward may rewrite the bf16 weight copies before the fan's product reads them."""
        dev, cur = wt.device, torch.cuda.current_stream(wt.device)
        if self.dctx is None:
            self.dctx = torch.empty((rows, self.n * dd), dtype=torch.bfloat16, device=dev)
            self.wcat = torch.empty((wt.shape[0], self.n * dd), dtype=torch.bfloat16, device=dev)  # no fill: the decoders write from several streams
        self.dctx.record_stream(cur)  # allocated on the first decoder's stream, written on each decoder's, read on the fan's
        self.wcat.record_stream(cur)
        self.wcat[:, i * dd:(i + 1) * dd].copy_(wt)
        self.written[i] = True
        return self.dctx[:, i * dd:(i + 1) * dd]


class ContextFanFn(torch.autograd.Function):
    """enc -> one small proxy tensor per decoder.  A decoder that is given a proxy returns no gradient for the encoder tokens
    themselves; it leaves d_ctx in its ContextFan slot and a (meaningless) gradient for the proxy, through which autograd
    orders this node's backward after every decoder's and synchronises their streams with this one."""

    @staticmethod
    def forward(ctx, enc, fan):
        ctx.fan, ctx.shape = fan, tuple(enc.shape)
        ctx.set_materialize_grads(False)
        return tuple(enc.new_empty(1) for _ in range(fan.n))  # never read: edges of the graph only

    @staticmethod
    def backward(ctx, *grads):
        fan, (B, seq, D) = ctx.fan, ctx.shape
        if fan.dctx is None:  # no decoder ran its backward
            return None, None
        dev = fan.dctx.device
        fan.dctx.record_stream(torch.cuda.current_stream(dev))
        fan.wcat.record_stream(torch.cuda.current_stream(dev))
        dd = fan.dctx.shape[1] // fan.n
        for i in range(fan.n):  # a decoder outside this backward pass contributes nothing
            if not fan.written[i]:
                fan.dctx[:, i * dd:(i + 1) * dd].zero_()
                fan.wcat[:, i * dd:(i + 1) * dd].zero_()
        d_enc = ops.gemm_nt(ops.EPI_F32, fan.dctx, fan.wcat, None)  # [M, n * Dd] x [D, n * Dd]^T
        fan.dctx = fan.wcat = None
        return d_enc.reshape(B, seq, D), None


class DecoderHeadFn(torch.autograd.Function):
    """multimae/output_adapters.py:341-353 of the reference: enc f32 [B,k+G,D] -> x f32 [B,ntok,Dd].
    tok: None, or this decoder's proxy from ContextFanFn (meta[8] = (ContextFan, index)): the gradient of enc is then left to
    that node."""

    @staticmethod
    def forward(ctx, enc, tok, keep32, restore32, meta, pos_table, mask_token, pc_w, pc_b, qn_w, qn_b, cn_w, cn_b, q_w, q_b,
                kv_w, kv_b, pr_w, pr_b, on_w, on_b, f1w, f1b, f2w, f2b, *task_embs):
        heads, k, G, T, ntok, q_start = meta[:6]
        ctx.fan = meta[8] if tok is not None and len(meta) > 8 else None
        B, seq, D = enc.shape
        Dd = pc_w.shape[0]
        d = Dd // heads
        # the bf16 copy of the encoder output, made once by the caller for all decoders (meta[6]) or here
        enc_b = meta[6] if len(meta) > 6 and meta[6] is not None else ops.cast_bf16(enc.contiguous().reshape(B * seq, D))
        wpc, wpct = bf16_weights(pc_w)
        ctxt = ops.gemm_nt(ops.EPI_BF16, enc_b, wpc, pc_b).reshape(B, seq, Dd)
        emb = ops.build_decoder_emb([None if e is None else e.detach().contiguous() for e in task_embs], pos_table.contiguous(), ntok, Dd)
        queries, context = ops.decoder_assemble_fwd(ctxt, mask_token.detach().contiguous(), emb, keep32, restore32, k, G, q_start, ntok)
        eps_q, eps_c, eps_o = meta[7] if len(meta) > 7 else (1e-6, 1e-6, 1e-6)  # of query_norm, context_norm, out_norm
        qn, qmean, qrstd = ops.layernorm_fwd(queries.reshape(B * ntok, Dd), qn_w, qn_b, eps_q)
        cn, cmean, crstd = ops.layernorm_fwd(context.reshape(B * seq, Dd), cn_w, cn_b, eps_c)
        wq, wqt = bf16_weights(q_w)
        wkv, wkvt = bf16_weights(kv_w)
        wpr, wprt = bf16_weights(pr_w)
        q = ops.gemm_nt(ops.EPI_BF16, qn, wq, q_b)
        kv = ops.gemm_nt(ops.EPI_BF16, cn, wkv, kv_b)
        o, lse = ops.attn_fwd(q, kv[:, :Dd], kv[:, Dd:], B, heads, ntok, seq, d, d ** -0.5)
        x = ops.gemm_nt(ops.EPI_F32, o, wpr, pr_b)  # the cross-attention output replaces the queries (no residual)
        y, s_mlp = _mlp_fwd(x, on_w, on_b, f1w, f1b, f2w, f2b, eps_o)
        ctx.saved = (s_mlp, o, lse, q, kv, qn, cn, queries, context, qmean, qrstd, cmean, crstd, enc_b,
                     wqt, wkvt, wprt, wpct, keep32, restore32, meta, (B, seq, D, Dd), (mask_token, task_embs),
                     (pc_w, pc_b, qn_w, qn_b, cn_w, cn_b, q_w, q_b, kv_w, kv_b, pr_w, pr_b))
        return y.reshape(B, ntok, Dd)

    @staticmethod
    def backward(ctx, g):
        (s_mlp, o, lse, q, kv, qn, cn, queries, context, qmean, qrstd, cmean, crstd, enc_b,
         wqt, wkvt, wprt, wpct, keep32, restore32, meta, dims, (mask_token, task_embs),
         (pc_w, pc_b, qn_w, qn_b, cn_w, cn_b, q_w, q_b, kv_w, kv_b, pr_w, pr_b)) = ctx.saved
        ctx.saved = None
        heads, k, G, T, ntok, q_start = meta[:6]
        B, seq, D, Dd = dims
        d = Dd // heads
        dx, g_mlp = _mlp_bwd(g.contiguous().reshape(B * ntok, Dd), s_mlp)
        dxb = _bf16(dx)
        d_o = ops.gemm_nt(ops.EPI_BF16, dxb, wprt, None)
        dwpr, dbpr = _lin_grads(dxb, o, pr_w, pr_b)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        ops.attn_bwd(q, kv[:, :Dd], kv[:, Dd:], o, d_o, lse, dq, dkv[:, :Dd], dkv[:, Dd:], B, heads, ntok, seq, d, d ** -0.5)
        dqn = ops.gemm_nt(ops.EPI_BF16, dq, wqt, None)
        dwq, dbq = _lin_grads(dq, qn, q_w, q_b)
        dcn = ops.gemm_nt(ops.EPI_BF16, dkv, wkvt, None)
        dwkv, dbkv = _lin_grads(dkv, cn, kv_w, kv_b)
        d_queries, dqnw, dqnb = _ln_bwd(dqn, queries.reshape(B * ntok, Dd), qn_w, qn_b, qmean, qrstd)
        d_context, dcnw, dcnb = _ln_bwd(dcn, context.reshape(B * seq, Dd), cn_w, cn_b, cmean, crstd)
        d_tok = None
        if ctx.fan is not None:  # d_ctx into this decoder's column block; the product with proj_context's weight is the fan's
            fan, slot = ctx.fan
            d_ctx2, d_mask, d_task = ops.decoder_assemble_bwd(d_queries.reshape(B, ntok, Dd), d_context.reshape(B, seq, Dd), keep32,
                                                              restore32, k, G, T, ntok, q_start, ntok,
                                                              d_ctx_out=fan.slot(slot, B * seq, Dd, wpct))
            d_enc, d_tok = None, g.new_empty(1)  # never read
        else:
            d_ctx, d_mask, d_task = ops.decoder_assemble_bwd(d_queries.reshape(B, ntok, Dd), d_context.reshape(B, seq, Dd), keep32,
                                                             restore32, k, G, T, ntok, q_start, ntok)
            d_ctx2 = d_ctx.reshape(B * seq, Dd)
            d_enc = ops.gemm_nt(ops.EPI_F32, d_ctx2, wpct, None).reshape(B, seq, D)
        dwpc, dbpc = _lin_grads(d_ctx2, enc_b, pc_w, pc_b)
        dnw, dnb, df1w, df1b, df2w, df2b = g_mlp
        d_mask = _deliver(mask_token, d_mask)
        d_task = [None if e is None else _deliver(e, d) for d, e in zip(d_task, task_embs)]
        return (d_enc, d_tok, None, None, None, None, d_mask, dwpc, dbpc, dqnw, dqnb, dcnw, dcnb, dwq, dbq, dwkv, dbkv, dwpr, dbpr,
                dnw, dnb, df1w, df1b, df2w, df2b, *d_task)


class ContextAssembleFn(torch.autograd.Function):
    """The first half of the decoder head as a node of its own (multimae/output_adapters.py:341-346 of the reference, body
    :230-314): proj_context, mask-token fill, un-shuffle, task / position embeddings -> (queries f32 [B,ntok,Dd], context
    f32 [B,k+G,Dd]).  Serves the decoder variants DecoderHeadFn does not fuse: no cross-attention (use_xattn=False), and
    queries that are not this task's slice of the context (use_task_queries=False, or a task that is not among the input
    domains): those are mask_token + position embedding (+ the task's embedding when the adapter has one), :280-293.

    meta = (k, G, T, ntok, q_start, own_queries, q_emb_index, enc_bf16 or None); q_emb_index = position of the query task's
    embedding in task_embs (the same parameter then serves context and queries and receives one summed gradient), -1 = the
    separate tensor q_emb (None when the adapter has no embedding for its task)."""

    @staticmethod
    def forward(ctx, enc, keep32, restore32, meta, pos_table, mask_token, pc_w, pc_b, q_emb, *task_embs):
        k, G, T, ntok, q_start, own_queries, q_emb_index, enc_b = meta
        B, seq, D = enc.shape
        Dd = pc_w.shape[0]
        if enc_b is None:
            enc_b = ops.cast_bf16(enc.contiguous().reshape(B * seq, D))
        wpc, wpct = bf16_weights(pc_w)
        ctxt = ops.gemm_nt(ops.EPI_BF16, enc_b, wpc, pc_b).reshape(B, seq, Dd)
        emb = ops.build_decoder_emb([None if e is None else e.detach().contiguous() for e in task_embs], pos_table.contiguous(), ntok, Dd)
        queries, context = ops.decoder_assemble_fwd(ctxt, mask_token.detach().contiguous(), emb, keep32, restore32, k, G,
                                                    0 if own_queries else q_start, ntok)
        if own_queries:
            row = mask_token.detach().reshape(1, Dd) + pos_table.detach().reshape(Dd, ntok).t()
            qe = task_embs[q_emb_index] if q_emb_index >= 0 else q_emb
            if qe is not None:
                row = row + qe.detach().reshape(1, Dd)
            queries = row.unsqueeze(0).expand(B, ntok, Dd).contiguous()
        ctx.saved = (enc_b, wpct, keep32, restore32, meta[:7], (B, seq, D, Dd), (mask_token, task_embs, q_emb, pc_w, pc_b))
        return queries, context

    @staticmethod
    def backward(ctx, d_queries, d_context):
        enc_b, wpct, keep32, restore32, (k, G, T, ntok, q_start, own_queries, q_emb_index), (B, seq, D, Dd), \
            (mask_token, task_embs, q_emb, pc_w, pc_b) = ctx.saved
        ctx.saved = None
        d_queries = d_queries.contiguous()
        dq_in = torch.zeros_like(d_queries) if own_queries else d_queries
        d_ctx, d_mask, d_task = ops.decoder_assemble_bwd(dq_in, d_context.contiguous(), keep32, restore32, k, G, T, ntok,
                                                         0 if own_queries else q_start, ntok)
        d_task = list(d_task)
        d_qemb = None
        if own_queries:  # every query row is the same parameter sum: its gradient is the sum over samples and positions
            dsum = d_queries.sum(dim=0).sum(dim=0)
            d_mask = d_mask + dsum.reshape(d_mask.shape)
            if q_emb_index >= 0:
                d_task[q_emb_index] = d_task[q_emb_index] + dsum.reshape(d_task[q_emb_index].shape)
            elif q_emb is not None:
                d_qemb = _deliver(q_emb, dsum.reshape(q_emb.shape))
        d_ctx2 = d_ctx.reshape(B * seq, Dd)
        d_enc = ops.gemm_nt(ops.EPI_F32, d_ctx2, wpct, None).reshape(B, seq, D)
        dwpc, dbpc = _lin_grads(d_ctx2, enc_b, pc_w, pc_b)
        d_mask = _deliver(mask_token, d_mask)
        d_task = [None if e is None else _deliver(e, d) for d, e in zip(d_task, task_embs)]
        return (d_enc, None, None, None, None, d_mask, dwpc, dbpc, d_qemb, *d_task)


class OutProjFn(torch.autograd.Function):
    """out_proj + un-patchify (multimae/output_adapters.py:361-372 of the reference): x f32 [B,ntok,Dd] -> bf16 [B,C,H,W]
    (f32 with out_f32: the un-rounded fp32 accumulators, for adapters listed in fp32_output_adapters)."""

    @staticmethod
    def forward(ctx, x, w, b, C_, P, NH, NW, out_f32=False):
        B, ntok, Dd = x.shape
        xb = ops.cast_bf16(x.contiguous().reshape(B * ntok, Dd))
        wb, wt = bf16_weights(w)
        img = ops.unpatchify_gemm(xb, wb, b, B, C_, P, NH, NW, out_f32=out_f32)
        ctx.saved = (xb, wt, (B, ntok, Dd, P), (w, b))
        return img

    @staticmethod
    def backward(ctx, dimg):
        xb, wt, (B, ntok, Dd, P), (w, b) = ctx.saved
        ctx.saved = None
        rows = ops.patchify_bf16(dimg.contiguous(), P)
        dx = ops.gemm_nt(ops.EPI_F32, rows, wt, None).reshape(B, ntok, Dd)
        dw, db = _lin_grads(rows, xb, w, b)
        return dx, dw, db, None, None, None, None, None


class MaskedCEFn(torch.autograd.Function):
    """Masked cross entropy (multimae/criterion.py:39-65 of the reference); value and gradient from one kernel pass."""

    @staticmethod
    def forward(ctx, logits, target, mask, patch, label_smoothing):
        loss, d_logits = ops.masked_ce(logits.contiguous(), target.contiguous(), mask, patch, label_smoothing, want_grad=True)
        ctx.save_for_backward(d_logits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_logits,) = ctx.saved_tensors
        if getattr(ctx, "unit_upstream", False):
            return d_logits, None, None, None, None
        return d_logits * g.to(d_logits.dtype), None, None, None, None


def mark_unit_upstream(losses) -> None:
    """Tell the graph nodes of these loss tensors (outputs of MaskedLossFn / MaskedCEFn) that the gradient arriving at them
    will be exactly 1, as for the terms of a plain sum that backward() is called on: their backward then hands out the
    gradient the forward kernel wrote instead of a scaled copy (an image-sized multiply per loss).  The mark sits on the
    node of one forward call, so it ends with that graph; other tensors and other graphs are unaffected."""
    for t in losses:
        node = getattr(t, "grad_fn", None)
        if isinstance(node, (MaskedLossFn._backward_cls, MaskedCEFn._backward_cls)):
            node.unit_upstream = True


class MaskedLossFn(torch.autograd.Function):
    """Masked MSE / L1 (multimae/criterion.py:96-130, :161-195 of the reference); value and gradient from one kernel pass."""

    @staticmethod
    def forward(ctx, pred, target, mask, kind, norm_pix, patch):
        loss, d_pred = ops.masked_loss(kind, norm_pix, pred.contiguous(), target.contiguous(), mask, patch, want_grad=True)
        ctx.save_for_backward(d_pred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_pred,) = ctx.saved_tensors
        if getattr(ctx, "unit_upstream", False):  # the caller vouched that g is exactly 1 (mark_unit_upstream)
            return d_pred, None, None, None, None, None
        return d_pred * g.to(d_pred.dtype), None, None, None, None, None


# ---------------------------------------------------------------------------------------------
# fp32 path: output adapters listed in fp32_output_adapters run outside the low-precision regime, as the reference runs them
# outside autocast (multimae/multimae.py:426-436): fp32 activations throughout, LayerNorm / GELU / attention in fp32
# arithmetic, and every nn.Linear product with fp32-faithful operands (hi + lo bf16 split of both sides, three products in
# one call of the bf16 MFMA GEMM over K-concatenated operands, fp32 accumulation; csrc/fp32.hip).
# ---------------------------------------------------------------------------------------------
_SPLIT_CACHE: Dict[int, tuple] = {}


def _kpad(n: int) -> int:
    """Block width of a split operand whose blocks run along the reduction index: a multiple of 128 when that costs little, so
    that three blocks make whole k-tiles of the 256x256 GEMM (2128 -> 2176); small widths stay as they are."""
    return round_up(n, 128) if n >= 512 else n


def _split_weights(w: torch.Tensor):
    """(W split along K for the forward product [N, 3 * kpad(K)], W^T split along K for the input gradient [K, 3 * kpad(N)])
    of an fp32 weight [N, K], pattern (hi, lo, hi); cached until the parameter changes."""
    key = id(w)
    ver = (w._version, _WEIGHT_EPOCH, w.data_ptr())
    ent = _SPLIT_CACHE.get(key)
    if ent is not None and ent[0] == ver and ent[1] is w:
        return ent[2], ent[3]
    w2 = w.detach().reshape(w.shape[0], -1)
    wk = ops.split_f32(w2, 1, block_cols=_kpad(w2.shape[1]))
    wtk = ops.split_f32(w2.t().contiguous(), 1, block_cols=_kpad(w2.shape[0]))
    _SPLIT_CACHE[key] = (ver, w, wk, wtk)
    return wk, wtk


class LinearF32Fn(torch.autograd.Function):
    """x W^T + b with fp32-faithful operands, x f32 [..., K] -> f32 [..., N].  Saved for the backward: the split of x (bf16
    [M, 3 * kpad(K)]), not x itself."""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.contiguous().reshape(-1, x.shape[-1])
        wk, wtk = _split_weights(w)
        xs = ops.split_f32(x2, 0, block_cols=_kpad(x2.shape[1]))  # [hi | hi | lo]
        y = ops.gemm_nt(ops.EPI_F32, xs, wk, None if b is None else b.detach())
        ctx.saved = (xs, wtk, (w, b), x.shape)
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, g):
        xs, wtk, (w, b), shape = ctx.saved
        ctx.saved = None
        K, N = shape[-1], w.shape[0]
        g2 = g.contiguous().reshape(-1, N)
        gs = ops.split_f32(g2, 0, block_cols=_kpad(N))  # [hi | hi | lo]
        dx = ops.gemm_nt(ops.EPI_F32, gs, wtk, None) if ctx.needs_input_grad[0] else None
        # dW = g^T x and db = colsum(g), the same three terms as three accumulating weight-gradient products over the column
        # blocks of the two splits: g_hi^T x_hi + g_hi^T x_lo + g_lo^T x_hi; the column sums of g_hi and g_lo ride along
        bg, bx = gs.shape[1] // 3, xs.shape[1] // 3
        g_hi, g_lo = gs[:, :N], gs[:, 2 * bg:2 * bg + N]
        x_hi, x_lo = xs[:, :K], xs[:, 2 * bx:2 * bx + K]
        dw = torch.empty((N, K), dtype=torch.float32, device=g.device)
        db = torch.empty((N,), dtype=torch.float32, device=g.device) if b is not None else None
        ops.gemm_tn(g_hi, x_hi, out=dw, accumulate=False, colsum_out=db, colsum_accumulate=False)
        ops.gemm_tn(g_hi, x_lo, out=dw, accumulate=True)
        ops.gemm_tn(g_lo, x_hi, out=dw, accumulate=True, colsum_out=db, colsum_accumulate=True)
        return (None if dx is None else dx.reshape(shape)), _deliver(w, dw.reshape(w.shape)), _deliver(b, db)


class LayerNormF32Fn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension, f32 in and out."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x2 = x.contiguous().reshape(-1, x.shape[-1])
        y, mean, rstd = ops.layernorm_fwd_f32(x2, w.detach(), b.detach(), eps)
        ctx.saved = (x2, mean, rstd, (w, b), x.shape)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, mean, rstd, (w, b), shape = ctx.saved
        ctx.saved = None
        dx, dg, db = ops.layernorm_bwd_f32(g.contiguous().reshape(x2.shape), x2, w.detach(), mean, rstd)
        return dx.reshape(shape), _deliver(w, dg), _deliver(b, db), None


class GeluF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        ctx.saved = xc
        return ops.gelu_f32(xc)

    @staticmethod
    def backward(ctx, g):
        x, ctx.saved = ctx.saved, None
        return ops.gelu_bwd_f32(x, g.contiguous())


class SelfAttnF32Fn(torch.autograd.Function):
    """softmax(q k^T scale) v over the packed projection qkv f32 [B*N, 3*H*d] -> f32 [B*N, H*d]."""

    @staticmethod
    def forward(ctx, qkv, B, H, N, d):
        D = H * d
        qkv = qkv.contiguous()
        o, lse = ops.attn_f32_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, N, N, d, d ** -0.5)
        ctx.saved = (qkv, o, lse, (B, H, N, d))
        return o

    @staticmethod
    def backward(ctx, g):
        qkv, o, lse, (B, H, N, d) = ctx.saved
        ctx.saved = None
        D = H * d
        dqkv = torch.empty_like(qkv)
        ops.attn_f32_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, g.contiguous(), lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                         B, H, N, N, d, d ** -0.5)
        return dqkv, None, None, None, None


class CrossAttnF32Fn(torch.autograd.Function):
    """softmax(q k^T scale) v with q f32 [B*Nq, H*d] and the packed context projection kv f32 [B*Nk, 2*H*d]."""

    @staticmethod
    def forward(ctx, q, kv, B, H, Nq, Nk, d):
        D = H * d
        q, kv = q.contiguous(), kv.contiguous()
        o, lse = ops.attn_f32_fwd(q, kv[:, :D], kv[:, D:], B, H, Nq, Nk, d, d ** -0.5)
        ctx.saved = (q, kv, o, lse, (B, H, Nq, Nk, d))
        return o

    @staticmethod
    def backward(ctx, g):
        q, kv, o, lse, (B, H, Nq, Nk, d) = ctx.saved
        ctx.saved = None
        D = H * d
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        ops.attn_f32_bwd(q, kv[:, :D], kv[:, D:], o, g.contiguous(), lse, dq, dkv[:, :D], dkv[:, D:], B, H, Nq, Nk, d, d ** -0.5)
        return dq, dkv, None, None, None, None, None


class AssembleF32Fn(torch.autograd.Function):
    """Queries / context assembly (multimae/output_adapters.py:230-314 of the reference) over an fp32 projected context:
    ctx f32 [B, k+G, Dd] -> (queries f32 [B, ntok, Dd], context f32 [B, k+G, Dd]); meta = (k, G, T, ntok, q_start)."""

    @staticmethod
    def forward(ctx_, ctx, keep32, restore32, meta, pos_table, mask_token, *task_embs):
        k, G, T, ntok, q_start = meta
        Dd = ctx.shape[-1]
        emb = ops.build_decoder_emb([None if e is None else e.detach().contiguous() for e in task_embs], pos_table.contiguous(), ntok, Dd)
        queries, context = ops.decoder_assemble_fwd(ctx.contiguous(), mask_token.detach().contiguous(), emb, keep32, restore32, k, G,
                                                    q_start, ntok)
        ctx_.saved = (keep32, restore32, meta, (mask_token, task_embs))
        return queries, context

    @staticmethod
    def backward(ctx_, d_queries, d_context):
        keep32, restore32, (k, G, T, ntok, q_start), (mask_token, task_embs) = ctx_.saved
        ctx_.saved = None
        d_ctx, d_mask, d_task = ops.decoder_assemble_bwd(d_queries.contiguous(), d_context.contiguous(), keep32, restore32, k, G, T, ntok,
                                                         q_start, ntok, f32=True)
        d_task = [None if e is None else _deliver(e, d) for d, e in zip(d_task, task_embs)]
        return (d_ctx, None, None, None, None, _deliver(mask_token, d_mask), *d_task)


def linear_f32(x, lin):
    return LinearF32Fn.apply(x, lin.weight, lin.bias)


def layer_norm_f32(norm, x):
    return LayerNormF32Fn.apply(x, norm.weight, norm.bias, norm.eps)


def block_f32(blk, x):
    """multimae_utils.Block in fp32 (the decoder blocks of an fp32 output adapter): x f32 [B, N, D]."""
    B, N, D = x.shape
    a, m = blk.attn, blk.mlp
    qkv = linear_f32(layer_norm_f32(blk.norm1, x), a.qkv)
    o = SelfAttnF32Fn.apply(qkv.reshape(B * N, 3 * D), B, a.num_heads, N, D // a.num_heads)
    x = x + linear_f32(o.reshape(B, N, D), a.proj)
    return x + linear_f32(GeluF32Fn.apply(linear_f32(layer_norm_f32(blk.norm2, x), m.fc1)), m.fc2)
