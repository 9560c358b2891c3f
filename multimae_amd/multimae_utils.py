"""Building blocks with the reference's class names and parameter layout (multimae/multimae_utils.py of the
reference); the arithmetic runs in the HIP kernels through multimae_amd.functional."""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import functional as Fn


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def build_2d_sincos_posemb(h, w, embed_dim=1024, temperature=10000.0):
    """Fixed 2-D sin-cos table [1, embed_dim, h, w] (reference: multimae/multimae_utils.py:29-49).

    Channel quarters are sin(a*w_j), cos(a*w_j), sin(b*w_j), cos(b*w_j) with w_j = temperature^(-j/(D/4)).
    The reference flattens a (w, h) grid and re-reads it as (h, w), so for the flat position m = row*w + col
    the coordinates are a = m // h and b = m % h; with h == w that makes `a` the image row."""
    if embed_dim % 4:
        raise ValueError("embed_dim must be divisible by 4 for the 2-D sin-cos embedding")
    quarter = embed_dim // 4
    freq = 1.0 / (temperature ** (torch.arange(quarter, dtype=torch.float32) / quarter))
    m = torch.arange(h * w)
    a = torch.div(m, h, rounding_mode="floor").float()[:, None] * freq[None, :]
    b = (m % h).float()[:, None] * freq[None, :]
    table = torch.cat([a.sin(), a.cos(), b.sin(), b.cos()], dim=1)
    return table.reshape(h, w, embed_dim).permute(2, 0, 1).unsqueeze(0).contiguous()


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    """In-place truncated normal through the inverse CDF (reference: multimae/multimae_utils.py:52-108)."""
    def cdf(v):
        return 0.5 * (1.0 + math.erf(v / math.sqrt(2.0)))
    with torch.no_grad():
        lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
        tensor.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_().mul_(std * math.sqrt(2.0)).add_(mean).clamp_(min=a, max=b)
    return tensor


def _attn_core_with_dropout(q, k, v, scale: float, p: float):
    """softmax(q k^T scale) -> dropout -> @ v with the probabilities materialised, q [B,H,Nq,d], k / v [B,H,Nk,d] fp32
    (reference: multimae/multimae_utils.py:200-204 and :240-244 with attn_drop > 0 in training mode).  The attention kernels
    keep the probabilities in registers and have no dropout, so this setting (0 in every pretraining configuration) runs
    the score and value products as plain batched library products around torch's softmax and dropout, between the same
    projection kernels; one F.dropout call on the [B,H,Nq,Nk] probabilities, as in the reference."""
    attn = (q @ k.transpose(-2, -1)) * scale
    attn = F.dropout(attn.softmax(dim=-1), p, True)
    return attn @ v


def drop_path(x, drop_prob: float = 0.0, training: bool = False):
    """Stochastic depth per sample (reference: multimae/multimae_utils.py:111-128, same draw: one torch.rand of shape
    [B, 1, ...] in x's dtype on x's device, binarised by floor(keep + u); survivors scaled by 1 / keep)."""
    if not drop_prob or not training:
        return x
    keep = 1.0 - drop_prob
    gate = keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)
    gate.floor_()
    return x.div(keep) * gate


class DropPath(nn.Module):
    """Drop paths (stochastic depth) per sample, applied to the residual branches of a block."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)

    def extra_repr(self) -> str:
        return "p={}".format(self.drop_prob)


class Mlp(nn.Module):
    """fc1 -> activation -> fc2 (reference: multimae/multimae_utils.py:145-169).  With nn.GELU (the reference's default,
    exact erf form) the activation is computed in the fc1 product's epilogue, and inside Block / SpatialOutputAdapter the
    chain runs fused with the LayerNorm and residual around it; any other act_layer runs as its torch module between the
    two products."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = None if act_layer is nn.GELU else act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = drop  # on the output of fc2 only (reference: multimae/multimae_utils.py:162-169)

    def forward(self, x):
        if self.act is None:
            y = Fn.MlpFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        else:
            h = self.act(Fn.LinearFn.apply(x, self.fc1.weight, self.fc1.bias))
            y = Fn.LinearFn.apply(h, self.fc2.weight, self.fc2.bias)
        return F.dropout(y, self.drop, True) if self.training and self.drop else y


class Attention(nn.Module):
    """qkv / proj parameters of multi-head self-attention (reference: multimae/multimae_utils.py:172-207)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.attn_drop, self.proj_drop = attn_drop, proj_drop

    def forward(self, x):
        if self.training and self.attn_drop:
            B, N, D = x.shape
            qkv = Fn.LinearFn.apply(x, self.qkv.weight, _bias_or_zeros(self.qkv))
            q, k, v = qkv.reshape(B, N, 3, self.num_heads, D // self.num_heads).permute(2, 0, 3, 1, 4).unbind(0)
            y = _attn_core_with_dropout(q, k, v, self.scale, self.attn_drop).transpose(1, 2).reshape(B, N, D)
            y = Fn.LinearFn.apply(y, self.proj.weight, self.proj.bias)
        else:
            y = Fn.AttentionFn.apply(x, self.num_heads, self.qkv.weight, _bias_or_zeros(self.qkv), self.proj.weight, self.proj.bias)
        return F.dropout(y, self.proj_drop, True) if self.training and self.proj_drop else y


class CrossAttention(nn.Module):
    """q / kv / proj parameters of multi-head cross-attention (reference: multimae/multimae_utils.py:210-247)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.attn_drop, self.proj_drop = attn_drop, proj_drop

    def forward(self, x, context):
        if self.training and self.attn_drop:
            B, N, D = x.shape
            M, H = context.shape[1], self.num_heads
            q = Fn.LinearFn.apply(x, self.q.weight, _bias_or_zeros(self.q)).reshape(B, N, H, D // H).permute(0, 2, 1, 3)
            k, v = Fn.LinearFn.apply(context, self.kv.weight, _bias_or_zeros(self.kv)).reshape(B, M, 2, H, D // H).permute(2, 0, 3, 1, 4).unbind(0)
            y = _attn_core_with_dropout(q, k, v, self.scale, self.attn_drop).transpose(1, 2).reshape(B, N, D)
            y = Fn.LinearFn.apply(y, self.proj.weight, self.proj.bias)
            return F.dropout(y, self.proj_drop, True) if self.proj_drop else y
        y = Fn.CrossAttentionFn.apply(x, context, self.num_heads, self.q.weight, _bias_or_zeros(self.q), self.kv.weight,
                                      _bias_or_zeros(self.kv), self.proj.weight, self.proj.bias)
        return F.dropout(y, self.proj_drop, True) if self.training and self.proj_drop else y


def _bias_or_zeros(lin: nn.Linear):
    if lin.bias is not None:
        return lin.bias
    if not hasattr(lin, "_mmae_zero_bias") or lin._mmae_zero_bias.device != lin.weight.device:
        lin._mmae_zero_bias = torch.zeros(lin.out_features, device=lin.weight.device)
    return lin._mmae_zero_bias


def layer_norm(norm: nn.LayerNorm, x):
    """norm(x) through the LayerNorm kernel as an autograd node of its own: f32 in, bf16 out (what Mlp / Attention /
    CrossAttention take as their operand)."""
    return Fn.LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps)


class Block(nn.Module):
    """x + attn(norm1(x)); x + mlp(norm2(x))  (reference: multimae/multimae_utils.py:250-285): one fused autograd node;
    with a drop rate > 0 in training mode, or a norm_layer other than an affine nn.LayerNorm, the same kernels as separate
    nodes with the dropout / stochastic depth / norm module of the reference between them."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        # the fused node normalises with the LayerNorm kernel (affine, over the last dimension, the module's eps); any
        # other norm module runs as itself between the separate nodes
        self._ln_kernel = all(isinstance(n, nn.LayerNorm) and n.elementwise_affine and n.bias is not None
                              and tuple(n.normalized_shape) == (dim,) for n in (self.norm1, self.norm2))

    def fused_node(self) -> bool:
        """True when forward() runs as the one fused autograd node (the pretraining configuration)."""
        a, m = self.attn, self.mlp
        return bool(self._ln_kernel and m.act is None
                    and not (self.training and (a.attn_drop or a.proj_drop or m.drop or isinstance(self.drop_path, DropPath))))

    def forward(self, x, chain=None):
        """chain: see functional.run_blocks (the encoder hands each block's first LayerNorm to the block before it)."""
        a, m = self.attn, self.mlp
        if not self._ln_kernel:
            x = x + self.drop_path(a(self.norm1(x)))
            return x + self.drop_path(m(self.norm2(x)))
        if not self.fused_node():
            x = x + self.drop_path(a(layer_norm(self.norm1, x)))
            return x + self.drop_path(m(layer_norm(self.norm2, x)))
        return Fn.BlockFn.apply(x, a.num_heads, self.norm1.weight, self.norm1.bias, a.qkv.weight, _bias_or_zeros(a.qkv),
                                a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias,
                                m.fc2.weight, m.fc2.bias, self.norm1.eps, self.norm2.eps, chain)
