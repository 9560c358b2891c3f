"""Output adapters with the reference's constructor / init / forward contract (multimae/output_adapters.py:43-374).
Only the cross-attention pretraining decoder is on the hot path; the finetuning heads of the reference
(linear / ConvNeXt / DPT) are out of scope here."""
from functools import partial
from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import functional as Fn
from .multimae_utils import Block, CrossAttention, Mlp, build_2d_sincos_posemb, layer_norm, pair, trunc_normal_


class SpatialOutputAdapter(nn.Module):
    """Cross-attention decoder for image-like outputs.

    proj_context -> (mask-token fill, un-shuffle, + task / position embeddings) -> queries = this task's slice
    -> cross-attention over the kept tokens (its output replaces the queries) -> MLP -> `depth` self-attention
    blocks -> out_proj -> image.  Parameter names and shapes follow the reference so state_dicts interchange.

    The pretraining configuration (task queries, cross-attention, no dropout) runs as one fused head node; the other
    settings of the reference run the same kernels as separate nodes: use_xattn=False (the queries go straight to the
    self-attention blocks, and the cross-attention / MLP modules do not exist, :135-148 and :349-355 there),
    use_task_queries=False or a task that is not an input domain (queries = mask token + position embedding + the task's
    embedding, :279-293), drop rates > 0 in training mode.  learnable_pos_emb=True is refused: the reference creates that
    table as [1, h, w, D] (:129-131) and its own forward then fails in the position-embedding add."""

    def __init__(self, num_channels: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens_enc: Optional[int] = None, dim_tokens: int = 256, depth: int = 0, learnable_pos_emb: int = False,
                 image_size: Union[int, Tuple[int]] = 224, mlp_ratio: int = 4.0, num_heads: int = 8, qkv_bias: bool = True,
                 drop_rate: float = 0.0, attn_drop_rate: float = 0.0, drop_path_rate: float = 0.0,
                 norm_layer: nn.Module = partial(nn.LayerNorm, eps=1e-6), use_task_queries: bool = True,
                 task: Optional[str] = None, context_tasks: Optional[list] = None, use_xattn: bool = True):
        super().__init__()
        self.num_channels = num_channels
        self.stride_level = stride_level
        self.patch_size_full = pair(patch_size_full)
        self.dim_tokens_enc = dim_tokens_enc
        self.dim_tokens = dim_tokens
        self.learnable_pos_emb = learnable_pos_emb
        self.image_size = pair(image_size)
        self.use_task_queries = use_task_queries
        self.task = task
        self.use_xattn = use_xattn
        self.num_heads = num_heads
        self.P_H = max(1, self.patch_size_full[0] // stride_level)
        self.P_W = max(1, self.patch_size_full[1] // stride_level)
        if learnable_pos_emb:
            raise NotImplementedError("learnable_pos_emb=True: the reference builds this table as [1, h, w, D] and its own forward "
                                      "fails on it; only the fixed sin-cos table is implemented")
        if self.P_H != self.P_W:
            raise NotImplementedError("square patches only")

        self.task_embeddings = None
        if context_tasks is not None:
            self.task_embeddings = nn.ParameterDict({t: nn.Parameter(torch.zeros(1, 1, dim_tokens)) for t in context_tasks})
            for emb in self.task_embeddings.values():
                trunc_normal_(emb, std=0.02)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, dim_tokens))
        h = self.image_size[0] // (stride_level * self.P_H)
        w = self.image_size[1] // (stride_level * self.P_W)
        self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h, w, dim_tokens), requires_grad=False)

        self._ln_kernel = True
        if use_xattn:
            self.decoder = CrossAttention(dim=dim_tokens, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop_rate,
                                          proj_drop=drop_rate)
            self.context_norm = norm_layer(dim_tokens)
            self.query_norm = norm_layer(dim_tokens)
            self.out_norm = norm_layer(dim_tokens)
            self.mlp = Mlp(in_features=dim_tokens, hidden_features=int(dim_tokens * mlp_ratio))
            # the LayerNorm kernel serves affine nn.LayerNorm over the last dimension; any other norm module runs as itself
            self._ln_kernel = all(isinstance(n, nn.LayerNorm) and n.elementwise_affine and n.bias is not None
                                  and tuple(n.normalized_shape) == (dim_tokens,)
                                  for n in (self.context_norm, self.query_norm, self.out_norm))
        if depth > 0:
            dpr = [r.item() for r in torch.linspace(0, drop_path_rate, depth)]  # stochastic depth decay rule
            self.decoder_transformer = nn.Sequential(*[
                Block(dim=dim_tokens, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                      attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer) for i in range(depth)])
        else:
            self.decoder_transformer = nn.Identity()
        self.dim_patch = num_channels * self.P_H * self.P_W
        self.out_proj = nn.Linear(dim_tokens, self.dim_patch)
        if dim_tokens_enc is not None:
            self.init(dim_tokens_enc=dim_tokens_enc)

    def init(self, dim_tokens_enc: int = 768):
        self.dim_tokens_enc = dim_tokens_enc
        self.proj_context = nn.Linear(dim_tokens_enc, self.dim_tokens)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_emb", "mask_token", "task_embeddings"}

    def _pos_table(self, nh: int, nw: int) -> Tensor:
        if tuple(self.pos_emb.shape[-2:]) == (nh, nw):
            return self.pos_emb
        return F.interpolate(self.pos_emb, size=(nh, nw), mode="bilinear", align_corners=False)

    def _unfused_head(self, encoder_tokens, keep32, restore32, dims, enc_b, tasks, embs, nh, nw):
        """Queries / context assembly, then (with use_xattn) cross-attention and MLP as separate autograd nodes."""
        k, G, T, ntok, q_start, own_queries = dims
        q_emb, q_emb_index = None, -1
        if own_queries and self.task_embeddings is not None and self.task in self.task_embeddings:
            if self.task in tasks:
                q_emb_index = tasks.index(self.task)
            else:
                q_emb = self.task_embeddings[self.task]
        meta = (k, G, T, ntok, q_start, own_queries, q_emb_index, enc_b)
        queries, context = Fn.ContextAssembleFn.apply(encoder_tokens, keep32, restore32, meta, self._pos_table(nh, nw), self.mask_token,
                                                      self.proj_context.weight, self.proj_context.bias, q_emb, *embs)
        if not self.use_xattn:
            return queries
        if not self._ln_kernel:
            x = self.decoder(self.query_norm(queries), self.context_norm(context))
            return x + self.mlp(self.out_norm(x))
        x = self.decoder(layer_norm(self.query_norm, queries), layer_norm(self.context_norm, context))
        return x + self.mlp(layer_norm(self.out_norm, x))

    def fused_head(self, input_info: Dict) -> bool:
        """True when forward() runs proj_context .. MLP as the one fused node (the pretraining configuration)."""
        own_queries = not self.use_task_queries or self.task not in input_info["tasks"]
        dropping = self.use_xattn and self.training and bool(self.decoder.attn_drop or self.decoder.proj_drop)
        return not (own_queries or not self.use_xattn or dropping or not self._ln_kernel)

    def _fp32_ready(self) -> bool:
        """The fp32 path serves the standard decoder (plain nn.Linear / nn.LayerNorm modules, GELU, head_dim 32 or 64)."""
        blocks = list(self.decoder_transformer) if isinstance(self.decoder_transformer, nn.Sequential) else []
        hd = self.dim_tokens // self.num_heads
        return (hd in (32, 64) and self.mlp.act is None and all(b.fused_node() for b in blocks)
                and all(b.attn.qkv.bias is not None for b in blocks) and self.decoder.q.bias is not None)

    def _forward_f32(self, encoder_tokens, keep32, restore32, dims, embs, nh, nw):
        """The decoder in fp32 end to end, as the reference runs the adapters listed in fp32_output_adapters outside autocast
        (multimae/multimae.py:426-436 there): fp32 activations, LayerNorm / GELU / softmax in fp32, every product with
        fp32-faithful operands (functional.LinearF32Fn).  Returns the prediction in fp32."""
        k, G, T, ntok, q_start = dims
        B, seq, _ = encoder_tokens.shape
        Dd, H = self.dim_tokens, self.num_heads
        ctx = Fn.linear_f32(encoder_tokens.float(), self.proj_context)
        queries, context = Fn.AssembleF32Fn.apply(ctx, keep32, restore32, (k, G, T, ntok, q_start), self._pos_table(nh, nw),
                                                  self.mask_token, *embs)
        d = self.decoder
        q = Fn.linear_f32(Fn.layer_norm_f32(self.query_norm, queries), d.q)
        kv = Fn.linear_f32(Fn.layer_norm_f32(self.context_norm, context), d.kv)
        o = Fn.CrossAttnF32Fn.apply(q.reshape(B * ntok, Dd), kv.reshape(B * seq, 2 * Dd), B, H, ntok, seq, Dd // H)
        x = Fn.linear_f32(o.reshape(B, ntok, Dd), d.proj)  # replaces the queries: no residual here (output_adapters.py:350-352)
        x = x + Fn.linear_f32(Fn.GeluF32Fn.apply(Fn.linear_f32(Fn.layer_norm_f32(self.out_norm, x), self.mlp.fc1)), self.mlp.fc2)
        if isinstance(self.decoder_transformer, nn.Sequential):
            for blk in self.decoder_transformer:
                x = Fn.block_f32(blk, x)
        y = Fn.linear_f32(x, self.out_proj)  # [B, nh * nw, C * P * P]
        C, P = self.num_channels, self.P_H
        # "b (nh nw) (c ph pw) -> b c (nh ph) (nw pw)" (output_adapters.py:364-372)
        return y.reshape(B, nh, nw, C, P, P).permute(0, 3, 1, 4, 2, 5).reshape(B, C, nh * P, nw * P)

    def forward(self, encoder_tokens: Tensor, input_info: Dict, ids_keep: Tensor, ids_restore: Tensor, fp32_output: bool = False):
        assert self.dim_tokens_enc is not None, "Need to call init(dim_tokens_enc) function first"
        H, W = input_info["image_size"]
        nh = H // (self.stride_level * self.P_H)
        nw = W // (self.stride_level * self.P_W)
        tasks = list(input_info["tasks"].keys())
        G = input_info.get("num_global_tokens", 0)
        k = encoder_tokens.shape[1] - G
        ntok = nh * nw
        for t in tasks:
            if input_info["tasks"][t]["num_tokens"] != ntok:
                raise NotImplementedError("all input domains must have the same token grid as this decoder")
        own_queries = not self.use_task_queries or self.task not in input_info["tasks"]
        q_start = 0 if own_queries else input_info["tasks"][self.task]["start_idx"]
        keep32 = input_info.get("_ids_keep_i32")
        restore32 = input_info.get("_ids_restore_i32")
        if keep32 is None or restore32 is None:
            keep32, restore32 = ids_keep.to(torch.int32).contiguous(), ids_restore.to(torch.int32).contiguous()
        embs = [self.task_embeddings[t] if (self.task_embeddings is not None and t in self.task_embeddings) else None for t in tasks]
        enc_b = input_info.get("_enc_bf16")  # made once for all decoders by MultiMAE.forward
        if enc_b is not None and (enc_b.shape[0] != encoder_tokens.shape[0] * encoder_tokens.shape[1] or enc_b.device != encoder_tokens.device):
            enc_b = None
        dropping = self.use_xattn and self.training and bool(self.decoder.attn_drop or self.decoder.proj_drop)
        if own_queries or not self.use_xattn or dropping or not self._ln_kernel:
            x = self._unfused_head(encoder_tokens, keep32, restore32, (k, G, len(tasks), ntok, q_start, own_queries), enc_b, tasks, embs, nh, nw)
            x = self.decoder_transformer(x)
            return Fn.OutProjFn.apply(x, self.out_proj.weight, self.out_proj.bias, self.num_channels, self.P_H, nh, nw, fp32_output)
        if fp32_output and self._fp32_ready():
            return self._forward_f32(encoder_tokens, keep32, restore32, (k, G, len(tasks), ntok, q_start), embs, nh, nw)
        d, m = self.decoder, self.mlp
        fan = (input_info.get("_ctx_fan") or {}).get(id(self))  # (ContextFan, slot, proxy) set up by MultiMAE.forward
        if fan is not None and fan[0].n * self.dim_tokens % 8:
            fan = None
        meta = (self.num_heads, k, G, len(tasks), ntok, q_start, enc_b, (self.query_norm.eps, self.context_norm.eps, self.out_norm.eps),
                None if fan is None else fan[:2])
        x = Fn.DecoderHeadFn.apply(
            encoder_tokens, None if fan is None else fan[2], keep32, restore32, meta, self._pos_table(nh, nw), self.mask_token,
            self.proj_context.weight, self.proj_context.bias, self.query_norm.weight, self.query_norm.bias,
            self.context_norm.weight, self.context_norm.bias, d.q.weight, _b(d.q), d.kv.weight, _b(d.kv), d.proj.weight, d.proj.bias,
            self.out_norm.weight, self.out_norm.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, *embs)
        x = self.decoder_transformer(x)
        return Fn.OutProjFn.apply(x, self.out_proj.weight, self.out_proj.bias, self.num_channels, self.P_H, nh, nw, fp32_output)


def _b(lin: nn.Linear):
    from .multimae_utils import _bias_or_zeros
    return _bias_or_zeros(lin)
