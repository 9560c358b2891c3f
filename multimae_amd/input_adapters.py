"""Input adapters with the reference's constructor / init / forward contract (multimae/input_adapters.py)."""
from typing import Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import nn

from . import functional as Fn
from .multimae_utils import build_2d_sincos_posemb, pair, trunc_normal_


class PatchedInputAdapter(nn.Module):
    """Image / feature map -> patch tokens + positional embedding (reference: multimae/input_adapters.py:27-137).

    Holds `proj` (Conv2d with kernel = stride = patch) and `pos_emb` [1, D, h, w] under the reference's names.
    Inside MultiMAE the projection is evaluated only for the tokens the mask keeps; calling the adapter directly
    projects every token (all-kept special case of the same kernels)."""

    def __init__(self, num_channels: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens: Optional[int] = None, sincos_pos_emb: bool = True, learnable_pos_emb: bool = False,
                 image_size: Union[int, Tuple[int]] = 224):
        super().__init__()
        self.num_channels = num_channels
        self.stride_level = stride_level
        self.patch_size_full = pair(patch_size_full)
        self.dim_tokens = dim_tokens
        self.sincos_pos_emb = sincos_pos_emb
        self.learnable_pos_emb = learnable_pos_emb
        self.image_size = pair(image_size)
        self.num_patches = (self.image_size[0] // self.patch_size_full[0]) * (self.image_size[1] // self.patch_size_full[1])
        self.P_H = max(1, self.patch_size_full[0] // stride_level)
        self.P_W = max(1, self.patch_size_full[1] // stride_level)
        if self.P_H != self.P_W:
            raise NotImplementedError("square patches only")
        if dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def init(self, dim_tokens: int = 768):
        self.dim_tokens = dim_tokens
        h = self.image_size[0] // (self.stride_level * self.P_H)
        w = self.image_size[1] // (self.stride_level * self.P_W)
        # as the reference (multimae/input_adapters.py:84-95): the sin-cos table is trained only with learnable_pos_emb, a
        # table that does not start from sin-cos always is
        if self.sincos_pos_emb:
            self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h, w, dim_tokens), requires_grad=bool(self.learnable_pos_emb))
        else:
            self.pos_emb = nn.Parameter(torch.zeros(1, dim_tokens, h, w))
            trunc_normal_(self.pos_emb, std=0.02)
        self.proj = nn.Conv2d(self.num_channels, dim_tokens, kernel_size=(self.P_H, self.P_W), stride=(self.P_H, self.P_W))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_emb"}

    def grid(self, H: int, W: int) -> Tuple[int, int]:
        assert self.dim_tokens is not None, "Need to call init(dim_tokens) function first"
        assert H % self.P_H == 0 and W % self.P_W == 0, \
            f"Image sizes {H}x{W} must be divisible by patch sizes {self.P_H}x{self.P_W}"
        return H // self.P_H, W // self.P_W

    def pos_table(self, nh: int, nw: int) -> torch.Tensor:
        """pos_emb resampled to the token grid (bicubic, as the reference; identity at the native size)."""
        if tuple(self.pos_emb.shape[-2:]) == (nh, nw):
            return self.pos_emb
        return F.interpolate(self.pos_emb, size=(nh, nw), mode="bicubic", align_corners=False)

    def learnable_pos_table(self, table: torch.Tensor):
        """`table` (the result of pos_table) when it takes a gradient, else None: the last argument group of EmbedFn."""
        return table if self.pos_emb.requires_grad and torch.is_grad_enabled() else None

    def forward(self, x):
        B, C, H, W = x.shape
        nh, nw = self.grid(H, W)
        keep = torch.arange(nh * nw, dtype=torch.int32, device=x.device).expand(B, -1).contiguous()
        empty_global = torch.zeros(1, 0, self.dim_tokens, device=x.device)
        table = self.pos_table(nh, nw)
        return Fn.EmbedFn.apply(keep, [self.P_H], (nh, nw), [table], empty_global, 1, x, self.proj.weight,
                                self.proj.bias, None, self.learnable_pos_table(table))

    class_table = None  # image domains have no class-embedding table


class SemSegInputAdapter(nn.Module):
    """Class map -> class embedding -> patch tokens + positional embedding.

    The reference fork imports this adapter (run_pretraining_multimae.py:42) and instantiates it as
    partial(SemSegInputAdapter, num_classes=133, dim_class_emb=64, interpolate_class_emb=False) with stride_level=4
    (:72-85) but does not contain its source; the input contract is an int64 map [B, H/stride, W/stride]
    (multimae/multimae.py:507-510).  Defined here from that call site: `class_emb` = Embedding(num_classes,
    dim_class_emb) (trunc-normal 0.02), `proj` = Conv2d(dim_class_emb, D, k=s=P) with P = patch_size_full // stride_level,
    frozen sin-cos `pos_emb`, bilinear resampling of pos_emb at other grid sizes."""

    def __init__(self, num_classes: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens: Optional[int] = None, sincos_pos_emb: bool = True, learnable_pos_emb: bool = False,
                 image_size: Union[int, Tuple[int]] = 224, dim_class_emb: int = 64, interpolate_class_emb: bool = False,
                 emb_padding_idx: Optional[int] = None):
        super().__init__()
        self.num_classes = num_classes
        self.stride_level = stride_level
        self.patch_size_full = pair(patch_size_full)
        self.dim_tokens = dim_tokens
        self.sincos_pos_emb = sincos_pos_emb
        self.learnable_pos_emb = learnable_pos_emb
        self.image_size = pair(image_size)
        self.dim_class_emb = dim_class_emb
        self.interpolate_class_emb = interpolate_class_emb
        self.emb_padding_idx = emb_padding_idx
        if emb_padding_idx is not None:
            self.num_classes += 1
        self.P_H = max(1, self.patch_size_full[0] // stride_level)
        self.P_W = max(1, self.patch_size_full[1] // stride_level)
        if interpolate_class_emb or emb_padding_idx is not None or self.P_H != self.P_W:
            raise NotImplementedError("the fused class-map embedding implements interpolate_class_emb=False, no padding index, "
                                      "square patches")
        if dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def init(self, dim_tokens: int = 768):
        self.dim_tokens = dim_tokens
        h = self.image_size[0] // (self.stride_level * self.P_H)
        w = self.image_size[1] // (self.stride_level * self.P_W)
        if self.sincos_pos_emb:
            self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h, w, dim_tokens), requires_grad=bool(self.learnable_pos_emb))
        else:
            self.pos_emb = nn.Parameter(torch.zeros(1, dim_tokens, h, w))
            trunc_normal_(self.pos_emb, std=0.02)
        self.class_emb = nn.Embedding(num_embeddings=self.num_classes, embedding_dim=self.dim_class_emb)
        trunc_normal_(self.class_emb.weight, std=0.02)
        self.proj = nn.Conv2d(self.dim_class_emb, dim_tokens, kernel_size=(self.P_H, self.P_W), stride=(self.P_H, self.P_W))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_emb"}

    @property
    def class_table(self):
        return self.class_emb.weight

    def grid(self, H: int, W: int) -> Tuple[int, int]:
        assert self.dim_tokens is not None, "Need to call init(dim_tokens) function first"
        assert H % self.P_H == 0 and W % self.P_W == 0, \
            f"Image sizes {H}x{W} must be divisible by patch sizes {self.P_H}x{self.P_W}"
        return H // self.P_H, W // self.P_W

    def pos_table(self, nh: int, nw: int) -> torch.Tensor:
        if tuple(self.pos_emb.shape[-2:]) == (nh, nw):
            return self.pos_emb
        return F.interpolate(self.pos_emb, size=(nh, nw), mode="bilinear", align_corners=False)

    learnable_pos_table = PatchedInputAdapter.learnable_pos_table

    def forward(self, x):
        B, H, W = x.shape
        nh, nw = self.grid(H, W)
        keep = torch.arange(nh * nw, dtype=torch.int32, device=x.device).expand(B, -1).contiguous()
        empty_global = torch.zeros(1, 0, self.dim_tokens, device=x.device)
        table = self.pos_table(nh, nw)
        return Fn.EmbedFn.apply(keep, [self.P_H], (nh, nw), [table], empty_global, 1, x.to(torch.int64),
                                self.proj.weight, self.proj.bias, self.class_emb.weight, self.learnable_pos_table(table))
