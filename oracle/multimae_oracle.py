"""CPU restatement of the MultiMAE pretraining step.  TEST INFRASTRUCTURE ONLY.

This file is the checker for the HIP path, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``multimae_amd/`` imports it.

It restates, as pure functions over a ``state_dict`` (reference key names), the
arithmetic of the reference files cited per function (paths relative to the
reference tree).  All randomness is injected: the Dirichlet sample and the
noise tensors are arguments, so the same inputs can be replayed through the
reference, this restatement and the HIP kernels.

Pinning: ``tests/golden/make_golden.py`` imports the reference ``multimae``
package, replays its RNG call order, and stores inputs/outputs as fixtures;
``tests/test_oracle_golden.py`` checks this file against them, and
``tests/test_oracle_vs_reference.py`` checks it live against the imported
reference when the reference tree is present.

Tie policy for the argsorts (the reference calls a non-stable argsort whose tie
order is implementation defined): stable, lower index first.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LN_EPS = 1e-6  # multimae/multimae.py:77, multimae/output_adapters.py:86


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
def make_cfg(
    in_domains: Dict[str, int],
    out_domains: Dict[str, Tuple[int, str]],
    dim: int = 768,
    depth: int = 12,
    heads: int = 12,
    dec_dim: int = 256,
    dec_depth: int = 2,
    dec_heads: int = 8,
    patch: int = 16,
    image: int = 224,
    num_global: int = 1,
    mlp_ratio: float = 4.0,
    stride: Optional[Dict[str, int]] = None,
    num_classes: Optional[Dict[str, int]] = None,
) -> dict:
    """in_domains: name -> channels (for a class-map domain: the class-embedding width).
    out_domains: name -> (channels, query task).  stride: name -> stride level of a domain (input or output; default 1).
    num_classes: name -> number of classes of a class-map input domain (its input is int64 [B, H/stride, W/stride])."""
    return dict(
        in_domains=dict(in_domains),
        out_domains={k: (int(c), str(t)) for k, (c, t) in out_domains.items()},
        dim=dim, depth=depth, heads=heads, dec_dim=dec_dim, dec_depth=dec_depth,
        dec_heads=dec_heads, patch=patch, image=image, num_global=num_global,
        mlp_ratio=mlp_ratio, stride=dict(stride or {}), num_classes=dict(num_classes or {}),
    )


def domain_patch(cfg: dict, name: str) -> int:
    """Patch edge in the domain's own pixels: max(1, patch_size_full // stride_level) (input_adapters.py:64-65)."""
    return max(1, cfg["patch"] // cfg.get("stride", {}).get(name, 1))


def vit_b_rgbd_cfg(extra_norm_pix: bool = True, image: int = 224) -> dict:
    """ViT-B/16 RGB-D with the reference's default decoders
    (pretrain_argparser.py:13-41, run_pretraining_multimae.py:57-71,118-130)."""
    out = {"rgb": (3, "rgb"), "depth": (1, "depth")}
    if extra_norm_pix:
        out["norm_rgb"] = (3, "rgb")
    return make_cfg({"rgb": 3, "depth": 1}, out, image=image)


def vit_b_rgbd_semseg_cfg(extra_norm_pix: bool = False, image: int = 224, num_classes: int = 133,
                          dim_class_emb: int = 64) -> dict:
    """ViT-B/16 RGB + depth + semantic segmentation (class map at stride 4, 133 COCO classes, 64-wide class embedding):
    run_pretraining_multimae.py:72-85 and cfgs/pretrain/multimae-b_98_rgb+-depth-semseg_1600e.yaml of the reference."""
    out = {"rgb": (3, "rgb"), "depth": (1, "depth"), "semseg": (num_classes, "semseg")}
    if extra_norm_pix:
        out["norm_rgb"] = (3, "rgb")
    return make_cfg({"rgb": 3, "depth": 1, "semseg": dim_class_emb}, out, image=image,
                    stride={"semseg": 4}, num_classes={"semseg": num_classes})


def small_semseg_cfg() -> dict:
    """small_cfg plus a class-map domain (9 classes, 8-wide class embedding, stride 4)."""
    out = {"rgb": (3, "rgb"), "depth": (1, "depth"), "semseg": (9, "semseg")}
    return make_cfg({"rgb": 3, "depth": 1, "semseg": 8}, out, dim=128, depth=2, heads=4, dec_dim=64,
                    dec_depth=1, dec_heads=2, patch=16, image=64, stride={"semseg": 4}, num_classes={"semseg": 9})


def tiny_cfg() -> dict:
    out = {"rgb": (3, "rgb"), "depth": (1, "depth"), "norm_rgb": (3, "rgb")}
    return make_cfg({"rgb": 3, "depth": 1}, out, dim=64, depth=2, heads=4, dec_dim=32,
                    dec_depth=1, dec_heads=4, patch=16, image=64)


def small_cfg() -> dict:
    """Smallest configuration the HIP kernels accept (head_dim 32 in encoder and decoders)."""
    out = {"rgb": (3, "rgb"), "depth": (1, "depth"), "norm_rgb": (3, "rgb")}
    return make_cfg({"rgb": 3, "depth": 1}, out, dim=128, depth=2, heads=4, dec_dim=64,
                    dec_depth=1, dec_heads=2, patch=16, image=64)


# --------------------------------------------------------------------------
# positional embedding (multimae/multimae_utils.py:29-49)
# --------------------------------------------------------------------------
def sincos_posemb(h: int, w: int, dim: int, temperature: float = 10000.0) -> Tensor:
    """[1, dim, h, w] table.  The reference builds a (w, h) 'ij' meshgrid, flattens it
    and then reads the flat index as row*w+col; for flat index m the first coordinate is
    m // h and the second m % h, so with h == w the sin/cos of the FIRST two channel
    quarters vary along image rows."""
    assert dim % 4 == 0
    q = dim // 4
    omega = 1.0 / (temperature ** (torch.arange(q, dtype=torch.float32) / q))
    m = torch.arange(h * w)
    a = (m // h).to(torch.float32)
    b = (m % h).to(torch.float32)
    oa = a[:, None] * omega[None, :]
    ob = b[:, None] * omega[None, :]
    tab = torch.cat([oa.sin(), oa.cos(), ob.sin(), ob.cos()], dim=1)  # [h*w, dim]
    return tab.reshape(h, w, dim).permute(2, 0, 1)[None].contiguous()


# --------------------------------------------------------------------------
# weight init (multimae/multimae.py:94-146, output_adapters.py:108-132)
# --------------------------------------------------------------------------
def _trunc_normal(shape, std, gen):
    # multimae/multimae_utils.py:52-87 with mean 0, a=-2, b=2
    lo = (1.0 + math.erf(-2.0 / std / math.sqrt(2.0))) / 2.0
    hi = (1.0 + math.erf(2.0 / std / math.sqrt(2.0))) / 2.0
    t = torch.empty(shape).uniform_(2 * lo - 1, 2 * hi - 1, generator=gen)
    return t.erfinv_().mul_(std * math.sqrt(2.0)).clamp_(-2.0, 2.0)


def _xavier(out_f, in_f, gen, fan_out=None):
    fo = out_f if fan_out is None else fan_out
    bound = math.sqrt(6.0 / float(fo + in_f))
    return torch.empty(out_f, in_f).uniform_(-bound, bound, generator=gen)


def init_state(cfg: dict, seed: int = 0) -> Dict[str, Tensor]:
    """Random state_dict with the reference's key names, shapes and init distributions
    (not its RNG stream: parity tests load identical weights into both sides)."""
    g = torch.Generator().manual_seed(seed)
    D, Dd, P = cfg["dim"], cfg["dec_dim"], cfg["patch"]
    hid, hid_d = int(D * cfg["mlp_ratio"]), int(Dd * cfg["mlp_ratio"])
    n = cfg["image"] // P
    sd: Dict[str, Tensor] = {}

    def block(prefix, d, hidden):
        sd[f"{prefix}.norm1.weight"] = torch.ones(d)
        sd[f"{prefix}.norm1.bias"] = torch.zeros(d)
        sd[f"{prefix}.attn.qkv.weight"] = _xavier(3 * d, d, g, fan_out=d)
        sd[f"{prefix}.attn.qkv.bias"] = torch.zeros(3 * d)
        sd[f"{prefix}.attn.proj.weight"] = _xavier(d, d, g)
        sd[f"{prefix}.attn.proj.bias"] = torch.zeros(d)
        sd[f"{prefix}.norm2.weight"] = torch.ones(d)
        sd[f"{prefix}.norm2.bias"] = torch.zeros(d)
        sd[f"{prefix}.mlp.fc1.weight"] = _xavier(hidden, d, g)
        sd[f"{prefix}.mlp.fc1.bias"] = torch.zeros(hidden)
        sd[f"{prefix}.mlp.fc2.weight"] = _xavier(d, hidden, g)
        sd[f"{prefix}.mlp.fc2.bias"] = torch.zeros(d)

    sd["global_tokens"] = _trunc_normal((1, cfg["num_global"], D), 0.02, g)
    for name, C in cfg["in_domains"].items():
        Pd = domain_patch(cfg, name)
        sd[f"input_adapters.{name}.pos_emb"] = sincos_posemb(n, n, D)
        if name in cfg.get("num_classes", {}):
            sd[f"input_adapters.{name}.class_emb.weight"] = _trunc_normal((cfg["num_classes"][name], C), 0.02, g)
        sd[f"input_adapters.{name}.proj.weight"] = _xavier(D, C * Pd * Pd, g).reshape(D, C, Pd, Pd)
        bound = 1.0 / math.sqrt(C * Pd * Pd)  # nn.Conv2d default bias init
        sd[f"input_adapters.{name}.proj.bias"] = torch.empty(D).uniform_(-bound, bound, generator=g)
    for i in range(cfg["depth"]):
        block(f"encoder.{i}", D, hid)
    for name, (C, _task) in cfg["out_domains"].items():
        p = f"output_adapters.{name}"
        Po = domain_patch(cfg, name)
        for t in cfg["in_domains"]:
            sd[f"{p}.task_embeddings.{t}"] = _trunc_normal((1, 1, Dd), 0.02, g)
        sd[f"{p}.mask_token"] = torch.zeros(1, 1, Dd)
        sd[f"{p}.pos_emb"] = sincos_posemb(n, n, Dd)
        sd[f"{p}.decoder.q.weight"] = _xavier(Dd, Dd, g)
        sd[f"{p}.decoder.q.bias"] = torch.zeros(Dd)
        sd[f"{p}.decoder.kv.weight"] = _xavier(2 * Dd, Dd, g, fan_out=Dd)
        sd[f"{p}.decoder.kv.bias"] = torch.zeros(2 * Dd)
        sd[f"{p}.decoder.proj.weight"] = _xavier(Dd, Dd, g)
        sd[f"{p}.decoder.proj.bias"] = torch.zeros(Dd)
        for nm in ("context_norm", "query_norm", "out_norm"):
            sd[f"{p}.{nm}.weight"] = torch.ones(Dd)
            sd[f"{p}.{nm}.bias"] = torch.zeros(Dd)
        sd[f"{p}.mlp.fc1.weight"] = _xavier(hid_d, Dd, g)
        sd[f"{p}.mlp.fc1.bias"] = torch.zeros(hid_d)
        sd[f"{p}.mlp.fc2.weight"] = _xavier(Dd, hid_d, g)
        sd[f"{p}.mlp.fc2.bias"] = torch.zeros(Dd)
        for j in range(cfg["dec_depth"]):
            block(f"{p}.decoder_transformer.{j}", Dd, hid_d)
        sd[f"{p}.out_proj.weight"] = _xavier(C * Po * Po, Dd, g)
        sd[f"{p}.out_proj.bias"] = torch.zeros(C * Po * Po)
        sd[f"{p}.proj_context.weight"] = _xavier(Dd, D, g)
        sd[f"{p}.proj_context.bias"] = torch.zeros(Dd)
    return sd


FROZEN_SUFFIX = ".pos_emb"  # requires_grad=False in the reference (input_adapters.py:88-90)


# --------------------------------------------------------------------------
# masking (multimae/multimae.py:191-253)
# --------------------------------------------------------------------------
def samples_per_task(task_sampling_dist: Tensor, num_encoded_tokens: int) -> Tensor:
    # multimae/multimae.py:218 : fp32 product, round half to even, int64
    return (task_sampling_dist * num_encoded_tokens).round().long()


def generate_masks(
    task_sampling_dist: Tensor,
    noises: Sequence[Tensor],
    noise_all: Tensor,
    num_encoded_tokens: int,
) -> Tuple[List[Tensor], Tensor, Tensor]:
    """task_sampling_dist [B,T] fp32 (the Dirichlet sample), noises[i] [B,N_i] fp32,
    noise_all [B,sum N_i] fp32 -> (masks per task int64 0=keep/1=masked,
    ids_keep [B,k] int64, ids_restore [B,sum N_i] int64)."""
    k_task = samples_per_task(task_sampling_dist, num_encoded_tokens)
    prelim = []
    for i, noise in enumerate(noises):
        order = torch.argsort(noise, dim=1, stable=True)  # :227-229
        # gather(arange, order) is `order` itself (:230-231); keep iff that INDEX < k_i (:233)
        prelim.append(torch.where(order < k_task[:, i : i + 1], 0, 1))
    mask_all = torch.cat(prelim, dim=1)
    key = mask_all + noise_all  # int64 + fp32 -> fp32 (:237)
    ids_shuffle = torch.argsort(key, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)  # inverse permutation
    ids_keep = ids_shuffle[:, :num_encoded_tokens]
    final = (ids_restore >= num_encoded_tokens).long()  # :241-245
    sizes = [n.shape[1] for n in noises]
    return list(torch.split(final, sizes, dim=1)), ids_keep, ids_restore


def draw_mask_randomness(B: int, sizes: Sequence[int], alphas, gen: Optional[torch.Generator] = None,
                         device="cpu"):
    """Draw (Dirichlet sample, per-task noise, global noise) in the reference's RNG call
    order (multimae/multimae.py:214-216,226,237).  Gamma draws normalised = Dirichlet."""
    T = len(sizes)
    a = torch.tensor([alphas] * T if isinstance(alphas, float) else list(alphas), dtype=torch.float32)
    conc = a.expand(B, T).contiguous()
    gam = torch._standard_gamma(conc.double(), generator=gen) if gen is not None else torch._standard_gamma(conc.double())
    p = (gam / gam.sum(-1, keepdim=True)).float()
    noises = [torch.rand(B, n, generator=gen) for n in sizes]
    noise_all = torch.rand(B, sum(sizes), generator=gen)
    return p.to(device), [n.to(device) for n in noises], noise_all.to(device)


# --------------------------------------------------------------------------
# transformer pieces (multimae/multimae_utils.py:145-285)
# --------------------------------------------------------------------------
def _ln(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{prefix}.weight"], sd[f"{prefix}.bias"], LN_EPS)


def _lin(x, sd, prefix):
    return F.linear(x, sd[f"{prefix}.weight"], sd.get(f"{prefix}.bias"))


def _mlp(x, sd, prefix):  # :162-169, exact (erf) GELU
    return _lin(F.gelu(_lin(x, sd, f"{prefix}.fc1")), sd, f"{prefix}.fc2")


def _softmax_attend(q, k, v, scale):
    att = (q @ k.transpose(-2, -1)) * scale
    att = att.softmax(dim=-1)
    return att @ v


def self_attention(x, sd, prefix, heads):  # :191-207
    B, N, C = x.shape
    d = C // heads
    qkv = _lin(x, sd, f"{prefix}.qkv").reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    o = _softmax_attend(qkv[0], qkv[1], qkv[2], d ** -0.5)
    return _lin(o.transpose(1, 2).reshape(B, N, C), sd, f"{prefix}.proj")


def cross_attention(x, ctx, sd, prefix, heads):  # :224-247
    B, N, C = x.shape
    M = ctx.shape[1]
    d = C // heads
    q = _lin(x, sd, f"{prefix}.q").reshape(B, N, heads, d).permute(0, 2, 1, 3)
    kv = _lin(ctx, sd, f"{prefix}.kv").reshape(B, M, 2, heads, d).permute(2, 0, 3, 1, 4)
    o = _softmax_attend(q, kv[0], kv[1], d ** -0.5)
    return _lin(o.transpose(1, 2).reshape(B, N, C), sd, f"{prefix}.proj")


def block(x, sd, prefix, heads):  # :282-285 (drop_path = 0)
    x = x + self_attention(_ln(x, sd, f"{prefix}.norm1"), sd, f"{prefix}.attn", heads)
    x = x + _mlp(_ln(x, sd, f"{prefix}.norm2"), sd, f"{prefix}.mlp")
    return x


# --------------------------------------------------------------------------
# input adapter (multimae/input_adapters.py:109-137)
# --------------------------------------------------------------------------
def patchify_cpp(img: Tensor, p: int) -> Tensor:
    """[B,C,H,W] -> [B, nh*nw, C*p*p] with the conv weight's (c, ph, pw) inner order."""
    B, C, H, W = img.shape
    nh, nw = H // p, W // p
    return img.reshape(B, C, nh, p, nw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, nh * nw, C * p * p)


def _resample(tab: Tensor, size: Tuple[int, int], mode: str) -> Tensor:
    if tuple(tab.shape[-2:]) == tuple(size):
        return tab  # identity resample (exact for bicubic and bilinear alike)
    return F.interpolate(tab, size=size, mode=mode, align_corners=False)


def semseg_input_adapter(class_map: Tensor, sd, prefix: str, p: int) -> Tensor:
    """Class map int64 [B,H,W] -> Embedding -> Conv2d(k=s=p) -> tokens + bilinear-resampled pos-emb.
    The reference fork does not contain this adapter (only its call site, run_pretraining_multimae.py:72-85, and the
    input contract at multimae/multimae.py:507-510); this follows the upstream MultiMAE definition the call-site arguments
    (num_classes, dim_class_emb, interpolate_class_emb=False) belong to.  PARITY UNPINNED."""
    B, H, W = class_map.shape
    assert H % p == 0 and W % p == 0
    emb = sd[f"{prefix}.class_emb.weight"][class_map].permute(0, 3, 1, 2)  # [B, C, H, W]
    w = sd[f"{prefix}.proj.weight"]
    tok = F.linear(patchify_cpp(emb, p), w.reshape(w.shape[0], -1), sd[f"{prefix}.proj.bias"])
    pos = _resample(sd[f"{prefix}.pos_emb"], (H // p, W // p), "bilinear")
    return tok + pos.flatten(2).transpose(1, 2)


def input_adapter(img: Tensor, sd, prefix: str, p: int) -> Tensor:
    B, C, H, W = img.shape
    assert H % p == 0 and W % p == 0
    w = sd[f"{prefix}.proj.weight"]
    tok = F.linear(patchify_cpp(img, p), w.reshape(w.shape[0], -1), sd[f"{prefix}.proj.bias"])
    pos = _resample(sd[f"{prefix}.pos_emb"], (H // p, W // p), "bicubic")
    return tok + pos.flatten(2).transpose(1, 2)


# --------------------------------------------------------------------------
# output adapter (multimae/output_adapters.py:195-374)
# --------------------------------------------------------------------------
def output_adapter(enc: Tensor, sd, prefix: str, cfg: dict, out_channels: int, task: str,
                   ids_keep: Tensor, ids_restore: Tensor, image_hw: Tuple[int, int], patch: Optional[int] = None) -> Tensor:
    """image_hw is the full-resolution image size; `patch` the output patch edge in the output's own pixels
    (patch_size_full // stride_level, output_adapters.py:101-102; token grid = image // patch_size_full, :332-334)."""
    B = enc.shape[0]
    G = cfg["num_global"]
    nh, nw = image_hw[0] // cfg["patch"], image_hw[1] // cfg["patch"]
    P = cfg["patch"] if patch is None else patch
    names = list(cfg["in_domains"].keys())
    n_tok = nh * nw
    total = n_tok * len(names)

    ctx = _lin(enc, sd, f"{prefix}.proj_context")  # :341
    ctx_ng = ctx[:, : ctx.shape[1] - G]
    Dd = ctx.shape[-1]
    fill = sd[f"{prefix}.mask_token"].expand(B, total - ctx_ng.shape[1], Dd)  # :251-259
    full = torch.cat([ctx_ng, fill], dim=1)
    full = torch.gather(full, 1, ids_restore[:, :, None].expand(-1, -1, Dd))  # :262-266
    pos = _resample(sd[f"{prefix}.pos_emb"], (nh, nw), "bilinear").flatten(2).transpose(1, 2)
    emb = torch.cat([sd[f"{prefix}.task_embeddings.{t}"] + pos for t in names], dim=1)  # :195-228
    full = full + emb
    assert task in names, "oracle covers the use_task_queries path of the pretraining configs"
    s = names.index(task) * n_tok
    queries = full[:, s : s + n_tok]  # :275-278
    kept = torch.gather(full, 1, ids_keep[:, :, None].expand(-1, -1, Dd))  # :296-300
    context = torch.cat([kept, ctx[:, ctx.shape[1] - G :]], dim=1)  # :303-310

    x = cross_attention(_ln(queries, sd, f"{prefix}.query_norm"), _ln(context, sd, f"{prefix}.context_norm"),
                        sd, f"{prefix}.decoder", cfg["dec_heads"])  # :350-352 (no residual)
    x = x + _mlp(_ln(x, sd, f"{prefix}.out_norm"), sd, f"{prefix}.mlp")  # :353
    for j in range(cfg["dec_depth"]):
        x = block(x, sd, f"{prefix}.decoder_transformer.{j}", cfg["dec_heads"])  # :358
    x = _lin(x, sd, f"{prefix}.out_proj")  # :361
    x = x.reshape(B, nh, nw, out_channels, P, P).permute(0, 3, 1, 4, 2, 5)  # :364-372
    return x.reshape(B, out_channels, nh * P, nw * P)


# --------------------------------------------------------------------------
# whole model (multimae/multimae.py:314-438)
# --------------------------------------------------------------------------
def forward(sd: Dict[str, Tensor], cfg: dict, x: Dict[str, Tensor], task_sampling_dist: Tensor,
            noises: Sequence[Tensor], noise_all: Tensor, num_encoded_tokens: int,
            return_encoder: bool = False):
    names = list(cfg["in_domains"].keys())
    first = x["rgb"] if "rgb" in x else x[names[0]]
    B, _, H, W = first.shape
    toks = [semseg_input_adapter(x[d], sd, f"input_adapters.{d}", domain_patch(cfg, d)) if d in cfg.get("num_classes", {})
            else input_adapter(x[d], sd, f"input_adapters.{d}", domain_patch(cfg, d)) for d in names]
    masks, ids_keep, ids_restore = generate_masks(task_sampling_dist, noises, noise_all, num_encoded_tokens)
    allt = torch.cat(toks, dim=1)
    D = allt.shape[-1]
    sel = torch.gather(allt, 1, ids_keep[:, :, None].expand(-1, -1, D))  # :398-402
    enc = torch.cat([sel, sd["global_tokens"].expand(B, -1, -1)], dim=1)  # :405-406
    for i in range(cfg["depth"]):
        enc = block(enc, sd, f"encoder.{i}", cfg["heads"])
    mask_d = {d: m for d, m in zip(names, masks)}
    if return_encoder:
        return enc, mask_d, ids_keep, ids_restore
    preds = {
        name: output_adapter(enc, sd, f"output_adapters.{name}", cfg, C, task, ids_keep, ids_restore, (H, W),
                             patch=domain_patch(cfg, name))
        for name, (C, task) in cfg["out_domains"].items()
    }
    return preds, mask_d


def multivit_forward(sd: Dict[str, Tensor], cfg: dict, x: Dict[str, Tensor], return_all_layers: bool = False):
    """MultiViT encoder pass: all tokens of the given domains + global tokens, no masking (multimae/multimae.py:501-563)."""
    names = [d for d in cfg["in_domains"] if d in x]
    B = next(iter(x.values())).shape[0]
    toks = [semseg_input_adapter(x[d], sd, f"input_adapters.{d}", domain_patch(cfg, d)) if d in cfg.get("num_classes", {})
            else input_adapter(x[d], sd, f"input_adapters.{d}", domain_patch(cfg, d)) for d in names]
    enc = torch.cat(toks + [sd["global_tokens"].expand(B, -1, -1)], dim=1)
    layers = []
    for i in range(cfg["depth"]):
        enc = block(enc, sd, f"encoder.{i}", cfg["heads"])
        layers.append(enc)
    return layers if return_all_layers else enc


# --------------------------------------------------------------------------
# losses (multimae/criterion.py)
# --------------------------------------------------------------------------
def _patch_standardise(target: Tensor, p: int) -> Tensor:
    # criterion.py:101-107 : per patch over (p1 p2 c), unbiased variance, eps 1e-6
    B, C, H, W = target.shape
    nh, nw = H // p, W // p
    t = target.reshape(B, C, nh, p, nw, p)
    mean = t.mean(dim=(1, 3, 5), keepdim=True)
    var = t.var(dim=(1, 3, 5), keepdim=True, unbiased=True)
    return ((t - mean) / torch.sqrt(var + 1e-6)).reshape(B, C, H, W)


def _masked_reduce(per_px: Tensor, mask: Optional[Tensor], p: int) -> Tensor:
    """per_px [B,H,W] (already averaged over channels); criterion.py:111-128."""
    if mask is None:
        return per_px.mean()
    if mask.sum() == 0:
        return torch.tensor(0).to(per_px.device)  # int64 zero, criterion.py:112-113
    B, H, W = per_px.shape
    nh, nw = H // p, W // p
    m = mask.reshape(B, nh, 1, nw, 1).float().expand(B, nh, p, nw, p).reshape(B, H, W)  # nearest
    per_sample = (per_px * m).flatten(1).sum(1) / m.flatten(1).sum(1)
    return per_sample.nanmean()


def masked_mse(pred, target, mask=None, patch_size=16, stride=1, norm_pix=False):  # :96-130
    p = patch_size // stride
    if norm_pix:
        target = _patch_standardise(target, p)
    err = (pred - target) ** 2
    if mask is None:
        return err.mean()
    return _masked_reduce(err.mean(dim=1), mask, p)


def masked_l1(pred, target, mask=None, patch_size=16, stride=1, norm_pix=False):  # :161-195
    p = patch_size // stride
    if norm_pix:
        target = _patch_standardise(target, p)
    err = (pred - target).abs()
    if mask is None:
        return err.mean()
    return _masked_reduce(err.mean(dim=1), mask, p)


def masked_ce(logits, target, mask=None, patch_size=16, stride=1, label_smoothing=0.0):  # :39-65
    p = patch_size // stride
    per_px = F.cross_entropy(logits, target, reduction="none", label_smoothing=label_smoothing)
    return _masked_reduce(per_px, mask, p)


# --------------------------------------------------------------------------
# truncated depth standardisation (run_pretraining_multimae.py:448-459)
# --------------------------------------------------------------------------
def standardize_depth(depth: Tensor) -> Tensor:
    B = depth.shape[0]
    s = torch.sort(depth.reshape(B, -1), dim=1)[0]
    n = s.shape[1]
    s = s[:, int(0.1 * n) : int(0.9 * n)]
    mean = s.mean(dim=1)[:, None, None, None]
    var = s.var(dim=1)[:, None, None, None]  # unbiased
    return (depth - mean) / torch.sqrt(var + 1e-6)


# --------------------------------------------------------------------------
# the step: forward, three masked losses, backward  (run_pretraining_multimae.py:465-498)
# --------------------------------------------------------------------------
def default_loss_table(cfg: dict):
    """name -> (fn, target domain, mask domain, kwargs) for the RGB-D pretraining configs
    (run_pretraining_multimae.py:57-71, 306-317: rgb MSE, depth L1, norm_rgb MSE norm_pix)."""
    tab = {}
    for name in cfg["out_domains"]:
        if name == "rgb":
            tab[name] = (masked_mse, "rgb", "rgb", dict(patch_size=cfg["patch"]))
        elif name == "depth":
            tab[name] = (masked_l1, "depth", "depth", dict(patch_size=cfg["patch"]))
        elif name == "norm_rgb":
            tab[name] = (masked_mse, "rgb", "rgb", dict(patch_size=cfg["patch"], norm_pix=True))
        elif name == "semseg":  # run_pretraining_multimae.py:72-85: masked CE at the class map's stride
            tab[name] = (masked_ce, "semseg", "semseg", dict(patch_size=cfg["patch"], stride=cfg["stride"].get("semseg", 1)))
        else:
            raise KeyError(name)
    return tab


def step_losses(sd, cfg, x, task_sampling_dist, noises, noise_all, num_encoded_tokens,
                autocast_dtype: Optional[torch.dtype] = None):
    """Forward + per-task losses. With autocast_dtype the model runs under
    torch.autocast (the reference's AMP semantics, :465) and predictions are cast to
    fp32 before the losses (:492-496)."""
    dev = next(iter(x.values())).device.type
    if autocast_dtype is not None:
        with torch.autocast(dev, dtype=autocast_dtype):
            preds, masks = forward(sd, cfg, x, task_sampling_dist, noises, noise_all, num_encoded_tokens)
    else:
        preds, masks = forward(sd, cfg, x, task_sampling_dist, noises, noise_all, num_encoded_tokens)
    losses = {}
    for name, (fn, tgt, mk, kw) in default_loss_table(cfg).items():
        losses[name] = fn(preds[name].float(), x[tgt], mask=masks[mk], **kw)
    return losses, preds, masks


def trainable(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    return {k: v for k, v in sd.items() if not k.endswith(FROZEN_SUFFIX)}


def step_grads(sd, cfg, x, task_sampling_dist, noises, noise_all, num_encoded_tokens,
               autocast_dtype: Optional[torch.dtype] = None):
    """Returns (losses, preds, masks, grads) with grads keyed like the state_dict."""
    work = {k: (v.detach().clone().requires_grad_(not k.endswith(FROZEN_SUFFIX))) for k, v in sd.items()}
    losses, preds, masks = step_losses(work, cfg, x, task_sampling_dist, noises, noise_all,
                                       num_encoded_tokens, autocast_dtype)
    total = sum(losses.values())
    names = [k for k in work if work[k].requires_grad]
    gs = torch.autograd.grad(total, [work[k] for k in names], allow_unused=True)
    grads = {k: g for k, g in zip(names, gs)}
    return ({k: v.detach() for k, v in losses.items()}, {k: v.detach() for k, v in preds.items()}, masks, grads)


class CpuTrainer:
    """fwd + losses + bwd + AdamW on the CPU; used as the bench's cpu_baseline ("port")."""

    def __init__(self, cfg, seed=0, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05):
        self.cfg = cfg
        sd = init_state(cfg, seed)
        self.sd = {k: v.requires_grad_(not k.endswith(FROZEN_SUFFIX)) for k, v in sd.items()}
        decay = [v for k, v in self.sd.items() if v.requires_grad and v.ndim > 1
                 and not any(s in k for s in ("global_tokens", "mask_token", "task_embeddings"))]
        no_decay = [v for k, v in self.sd.items() if v.requires_grad and not any(v is d for d in decay)]
        self.opt = torch.optim.AdamW([{"params": decay, "weight_decay": weight_decay},
                                      {"params": no_decay, "weight_decay": 0.0}], lr=lr, betas=betas)

    def step(self, x, num_encoded_tokens, gen=None):
        B = next(iter(x.values())).shape[0]
        n = (self.cfg["image"] // self.cfg["patch"]) ** 2
        p, noises, noise_all = draw_mask_randomness(B, [n] * len(self.cfg["in_domains"]), 1.0, gen)
        losses, _, _ = step_losses(self.sd, self.cfg, x, p, noises, noise_all, num_encoded_tokens)
        total = sum(losses.values())
        self.opt.zero_grad(set_to_none=True)
        total.backward()
        self.opt.step()
        return float(total)
