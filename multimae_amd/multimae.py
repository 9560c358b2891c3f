"""MultiMAE with the reference's module API (multimae/multimae.py), evaluated by the HIP kernels.

Order of work in forward differs from the reference where it saves work without changing results: the masks
depend only on shapes and noise, so they are drawn first and the patch projection is evaluated for the kept
tokens only."""
import itertools
import math
from collections import OrderedDict
from functools import partial
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor, nn
from torch.distributions.dirichlet import Dirichlet

from . import functional as Fn
from . import ops
from .input_adapters import PatchedInputAdapter, SemSegInputAdapter
from .multimae_utils import Block, trunc_normal_
from .output_adapters import SpatialOutputAdapter

__all__ = ["MultiMAE", "pretrain_multimae_base", "pretrain_multimae_large", "create_model"]

import os as _os
_SHARED_ENC_CAST = _os.environ.get("MMAE_SHARED_ENC_CAST", "1") != "0"  # tuning switch
_CTX_FAN = _os.environ.get("MMAE_CTX_FAN", "1") != "0"  # tuning switch: one product for the decoders' gradient of the encoder output


class MultiMAE(nn.Module):
    """Multi-modal masked autoencoder: per-domain patch adapters, Dirichlet-budgeted random masking, a ViT encoder
    over the kept tokens plus global tokens, and one cross-attention decoder per output domain."""

    def __init__(self, input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]],
                 num_global_tokens: int = 1, dim_tokens: int = 768, depth: int = 12, num_heads: int = 12,
                 mlp_ratio: float = 4.0, qkv_bias: bool = True, drop_rate: float = 0.0, attn_drop_rate: float = 0.0,
                 drop_path_rate: float = 0.0, norm_layer: nn.Module = partial(nn.LayerNorm, eps=1e-6)):
        super().__init__()
        for adapter in input_adapters.values():
            adapter.init(dim_tokens=dim_tokens)
        self.input_adapters = nn.ModuleDict(input_adapters)
        if output_adapters is not None:
            for adapter in output_adapters.values():
                adapter.init(dim_tokens_enc=dim_tokens)
            self.output_adapters = nn.ModuleDict(output_adapters)
        else:
            self.output_adapters = None

        self.num_global_tokens = num_global_tokens
        self.global_tokens = nn.Parameter(torch.zeros(1, num_global_tokens, dim_tokens))
        trunc_normal_(self.global_tokens, std=0.02)

        dpr = [r.item() for r in torch.linspace(0, drop_path_rate, depth)]  # stochastic depth decay rule (reference: multimae/multimae.py:98-110)
        self.encoder = nn.Sequential(*[
            Block(dim=dim_tokens, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer) for i in range(depth)])
        self._reset_parameters()
        self._last_ids_i32 = None
        self._concentrations = {}  # (device, alphas) -> Dirichlet concentrations on the device

    # -- initialisation (reference: multimae/multimae.py:117-146) ------------------------------------
    def _reset_parameters(self):
        def base(m):
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        self.apply(base)
        for name, m in self.named_modules():
            if isinstance(m, nn.Linear):
                # packed projections: each of the 3 (qkv) / 2 (kv) blocks gets its own xavier bound
                parts = 3 if "qkv" in name else (2 if "kv" in name else 0)
                if parts:
                    bound = math.sqrt(6.0 / float(m.weight.shape[0] // parts + m.weight.shape[1]))
                    nn.init.uniform_(m.weight, -bound, bound)
            if isinstance(m, nn.Conv2d) and ".proj" in name:
                nn.init.xavier_uniform_(m.weight.data.view(m.weight.shape[0], -1))

    @torch.jit.ignore
    def no_weight_decay(self):
        skip = {"global_tokens"}
        for group, adapters in (("input_adapters", self.input_adapters), ("output_adapters", self.output_adapters or {})):
            for task, adapter in adapters.items():
                if hasattr(adapter, "no_weight_decay"):
                    skip |= {f"{group}.{task}.{n}" for n in adapter.no_weight_decay()}
        return skip

    # -- masking (reference: multimae/multimae.py:169-253) ---------------------------------------------
    def sample_alphas(self, B: int, n_tasks: int, alphas: float = 1.0, eps: float = 1e-5):
        """Per sample, pick a non-empty subset of tasks uniformly; Dirichlet concentration = alphas on the subset, eps elsewhere."""
        choices = torch.Tensor([list(c) for c in itertools.product([0, 1], repeat=n_tasks)][1:])
        pick = torch.randint(0, len(choices), (B,))
        return torch.index_select(choices, 0, pick) * torch.tensor(alphas) + eps

    def generate_random_masks(self, input_tokens: Dict[str, torch.Tensor], num_encoded_tokens: int,
                              alphas: Union[float, List[float]] = 1.0, sample_tasks_uniformly: bool = False,
                              mask_randomness=None):
        """Returns (task_masks {domain: [B,N_d] int64, 0 keep / 1 masked}, ids_keep [B,k] int64, ids_restore [B,sum N] int64).

        Only shapes and device of `input_tokens` are used.  `mask_randomness` = (task_sampling_dist [B,T],
        [noise_d [B,N_d]], noise_all [B,sum N]) replaces the draws (tests replay the reference's samples)."""
        first = next(iter(input_tokens.values()))
        B, device = first.shape[0], first.device
        sizes = [t.shape[1] for t in input_tokens.values()]
        if mask_randomness is not None:
            p, noises, noise_all = mask_randomness
            p, noise_all = p.to(device, torch.float32), noise_all.to(device, torch.float32)
            noise = torch.cat([n.to(device, torch.float32) for n in noises], dim=1)
        else:
            al = [alphas] * len(sizes) if isinstance(alphas, float) else alphas
            if sample_tasks_uniformly:
                p = Dirichlet(self.sample_alphas(B, len(sizes), alphas=al)).sample().to(device)
            else:
                # same draw as Dirichlet(concentration).sample((B,)) (reference: multimae/multimae.py:214-216) without the
                # per-call host->device copy of the concentrations and the distribution's argument check, each of
                # which makes the host wait for the device once per step
                if any(not a > 0 for a in al):
                    raise ValueError(f"alphas must be positive, got {al}")
                key = (str(device), tuple(float(a) for a in al))
                conc = self._concentrations.get(key)
                if conc is None:
                    conc = self._concentrations[key] = torch.tensor(key[1], dtype=torch.float32, device=device)
                p = torch._sample_dirichlet(conc.expand(B, len(sizes)))
            noise = torch.cat([torch.rand(B, n, device=device) for n in sizes], dim=1)
            noise_all = torch.rand(B, sum(sizes), device=device)
        mask, ids_keep, ids_restore, keep32, restore32 = ops.mask_generate(
            p.contiguous(), noise.contiguous(), noise_all.contiguous(), sizes, int(num_encoded_tokens))
        self._last_ids_i32 = (keep32, restore32)
        task_masks = {d: m for d, m in zip(input_tokens.keys(), torch.split(mask, sizes, dim=1))}
        return task_masks, ids_keep, ids_restore

    @staticmethod
    def make_mask(N_H, N_W, xy_idxs, full_tasks=[], indicate_visible=True, flatten=True, device="cuda"):
        """Masks from lists of visible (x, y) token coordinates per task (reference: multimae/multimae.py:255-285)."""
        masks = {}
        for k, v in xy_idxs.items():
            m = torch.ones(N_H, N_W, device=device)
            idx = torch.LongTensor(v)
            if len(idx) > 0:
                m[idx[:, 1], idx[:, 0]] = 0
            if k in full_tasks:
                m[:] = 0
            if not indicate_visible:
                m = 1 - m
            masks[k] = m.flatten().unsqueeze(0) if flatten else m
        return masks

    def generate_input_info(self, input_task_tokens: Dict[str, Tensor], image_size: Tuple[int, int]) -> Dict:
        """start/end index of every domain in the concatenated token sequence (reference: multimae/multimae.py:287-312)."""
        info = OrderedDict()
        info["tasks"] = {}
        start = 0
        for domain, tensor in input_task_tokens.items():
            n = int(tensor.shape[1])
            info["tasks"][domain] = {"num_tokens": n, "has_2d_posemb": True, "start_idx": start, "end_idx": start + n}
            start += n
        info["image_size"] = image_size
        info["num_task_tokens"] = start
        info["num_global_tokens"] = self.num_global_tokens
        return info

    # -- forward (reference: multimae/multimae.py:314-438) ---------------------------------------------
    def forward(self, x: Union[Dict[str, torch.Tensor], torch.Tensor], mask_inputs: bool = True,
                task_masks: Dict[str, torch.Tensor] = None, num_encoded_tokens: int = 128,
                alphas: Union[float, List[float]] = 1.0, sample_tasks_uniformly: bool = False,
                fp32_output_adapters: List[str] = [], mask_randomness=None):
        x = {"rgb": x} if isinstance(x, torch.Tensor) else x
        domains = [d for d in x if d in self.input_adapters]
        adapters = [self.input_adapters[d] for d in domains]
        # full-resolution image size (reference: multimae/multimae.py:345-352, :507-510): from rgb when given, else from
        # the first input scaled by its stride level
        ref_d = "rgb" if "rgb" in x else domains[0]
        B = x[ref_d].shape[0]
        H, W = (s * getattr(self.input_adapters[ref_d], "stride_level", 1) if ref_d != "rgb" else s for s in x[ref_d].shape[-2:])
        nh = nw = None
        for d, ad in zip(domains, adapters):
            if not isinstance(ad, (PatchedInputAdapter, SemSegInputAdapter)):
                raise NotImplementedError(f"input adapter of '{d}': not one of the adapters the fused patch-embed kernel serves")
            g = ad.grid(*x[d].shape[-2:])
            if nh is None:
                nh, nw = g
            elif g != (nh, nw):
                raise NotImplementedError("all input domains must map to the same token grid")
        device = x[domains[0]].device
        Fn.prepare_streams(device, len(self.output_adapters) if self.output_adapters else 0)
        specs = OrderedDict((d, torch.empty((B, nh * nw, 0), device=device)) for d in domains)
        input_info = self.generate_input_info(specs, image_size=(H, W))

        if not mask_inputs:
            num_encoded_tokens = nh * nw * len(domains)
        elif num_encoded_tokens is None:
            num_encoded_tokens = self.num_encoded_tokens

        if task_masks is None:
            task_masks, ids_keep, ids_restore = self.generate_random_masks(
                specs, num_encoded_tokens, alphas=alphas, sample_tasks_uniformly=sample_tasks_uniformly,
                mask_randomness=mask_randomness)
            keep32, restore32 = self._last_ids_i32
        else:
            mask_all = torch.cat([task_masks[d] for d in domains], dim=1)
            ids_shuffle = torch.argsort(mask_all, dim=1, stable=True)
            ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
            kept = (mask_all == 0).sum(dim=1)
            if not bool((kept == kept[0]).all()):
                raise ValueError("given task_masks must keep the same number of tokens in every sample")
            ids_keep = ids_shuffle[:, : int(kept[0])]
            keep32, restore32 = ids_keep.to(torch.int32).contiguous(), ids_restore.to(torch.int32).contiguous()
        input_info["_ids_keep_i32"], input_info["_ids_restore_i32"] = keep32, restore32

        tables = [ad.pos_table(nh, nw) for ad in adapters]
        tokens = Fn.EmbedFn.apply(keep32, [ad.P_H for ad in adapters], (nh, nw), tables,
                                  self.global_tokens, len(domains), *[x[d] for d in domains], *[ad.proj.weight for ad in adapters],
                                  *[ad.proj.bias for ad in adapters], *[ad.class_table for ad in adapters],
                                  *[ad.learnable_pos_table(t) for ad, t in zip(adapters, tables)])
        encoder_tokens = Fn.run_blocks(self.encoder, tokens)
        if self.output_adapters is None:
            return encoder_tokens, task_masks
        # Adapters listed in fp32_output_adapters run in fp32 end to end, as the reference runs them outside autocast
        # (multimae/multimae.py:426-436): SpatialOutputAdapter._forward_f32.
        def decode(d):
            return self.output_adapters[d](encoder_tokens=encoder_tokens, input_info=input_info, ids_keep=ids_keep,
                                           ids_restore=ids_restore, fp32_output=d in fp32_output_adapters)
        if encoder_tokens.is_cuda and len(self.output_adapters) > 1 and _SHARED_ENC_CAST:
            # every decoder projects the same encoder output: one bf16 copy for all of them, made before they are dealt over
            # the streams (DecoderHeadFn would otherwise cast it once per decoder)
            Bq, Nq, Dq = encoder_tokens.shape
            input_info["_enc_bf16"] = ops.cast_bf16(encoder_tokens.detach().contiguous().reshape(Bq * Nq, Dq))
        # decoders that run the fused head share one product for the gradient of the encoder output (Fn.ContextFan)
        if encoder_tokens.is_cuda and encoder_tokens.requires_grad and torch.is_grad_enabled() and _CTX_FAN:
            heads = [a for d, a in self.output_adapters.items()
                     if hasattr(a, "fused_head") and a.fused_head(input_info) and d not in fp32_output_adapters]
            if len(heads) > 1 and len({a.dim_tokens for a in heads}) == 1:
                fan = Fn.ContextFan(len(heads))
                proxies = Fn.ContextFanFn.apply(encoder_tokens, fan)
                input_info["_ctx_fan"] = {id(a): (fan, i, proxies[i]) for i, a in enumerate(heads)}
        lanes = Fn.decoder_lanes(encoder_tokens.device, len(self.output_adapters)) if encoder_tokens.is_cuda else None
        if lanes is not None:
            # the decoders are independent until the losses: dealt over the main stream and the decoder streams; same
            # kernels, same order per decoder, so the results do not change
            main = torch.cuda.current_stream(encoder_tokens.device)
            preds, where = {}, {}
            for st in lanes:  # before any decoder is queued: a later wait would put that lane behind the decoders on main
                if st != main:
                    st.wait_stream(main)
                    if "_enc_bf16" in input_info:
                        input_info["_enc_bf16"].record_stream(st)  # allocated on the main stream, read on this one
            for i, d in enumerate(self.output_adapters):
                st = where[d] = lanes[i % len(lanes)]
                with torch.cuda.stream(st):
                    preds[d] = decode(d)
            for d, st in where.items():
                if st != main:
                    main.wait_stream(st)
                    preds[d].record_stream(main)
            return preds, task_masks
        return {d: decode(d) for d in self.output_adapters}, task_masks


class MultiViT(MultiMAE):
    """MultiMAE without masking: every token of every given domain plus the global tokens goes through the encoder
    (reference: multimae/multimae.py:481-574).  With output_adapters=None it returns the encoder tokens [B, sum N + G, D]
    (or the list of every layer's tokens with return_all_layers); output adapters, when given, are called as
    adapter(encoder_tokens=..., input_info=...) like the reference's finetuning heads."""

    def process_input(self, x):
        x = {"rgb": x} if isinstance(x, torch.Tensor) else x
        domains = [d for d in x if d in self.input_adapters]
        adapters = [self.input_adapters[d] for d in domains]
        ref_d = "rgb" if "rgb" in x else ("semseg" if "semseg" in x else domains[0])  # reference :505-514
        B = x[ref_d].shape[0]
        H, W = (s * (getattr(self.input_adapters[ref_d], "stride_level", 1) if ref_d == "semseg" else 1) for s in x[ref_d].shape[-2:])
        grids = [ad.grid(*x[d].shape[-2:]) for d, ad in zip(domains, adapters)]
        if any(g != grids[0] for g in grids):
            raise NotImplementedError("all input domains must map to the same token grid")
        nh, nw = grids[0]
        device = x[domains[0]].device
        specs = OrderedDict((d, torch.empty((B, nh * nw, 0), device=device)) for d in domains)
        input_info = self.generate_input_info(specs, image_size=(H, W))
        keep32 = torch.arange(nh * nw * len(domains), dtype=torch.int32, device=device).expand(B, -1).contiguous()
        tables = [ad.pos_table(nh, nw) for ad in adapters]
        tokens = Fn.EmbedFn.apply(keep32, [ad.P_H for ad in adapters], (nh, nw), tables,
                                  self.global_tokens, len(domains), *[x[d] for d in domains], *[ad.proj.weight for ad in adapters],
                                  *[ad.proj.bias for ad in adapters], *[ad.class_table for ad in adapters],
                                  *[ad.learnable_pos_table(t) for ad, t in zip(adapters, tables)])
        return tokens, input_info

    def forward(self, x: Union[Dict[str, torch.Tensor], torch.Tensor], return_all_layers=False, **kwargs):
        input_tokens, input_info = self.process_input(x)
        if not return_all_layers:
            encoder_tokens = self.encoder(input_tokens)
        else:
            encoder_tokens, tokens = [], input_tokens
            for block in self.encoder:
                tokens = block(tokens)
                encoder_tokens.append(tokens)
        if self.output_adapters is None:
            return encoder_tokens
        return {d: self.output_adapters[d](encoder_tokens=encoder_tokens, input_info=input_info) for d in self.output_adapters}


_MODEL_REGISTRY = {}


def register_model(fn):
    _MODEL_REGISTRY[fn.__name__] = fn
    return fn


def create_model(model_name: str, **kwargs):
    """Name -> factory lookup used by the reference's training scripts (utils/model_builder.py:29-79 there)."""
    if model_name not in _MODEL_REGISTRY:
        raise RuntimeError(f"Unknown model ({model_name})")
    return _MODEL_REGISTRY[model_name](**kwargs)


@register_model
def pretrain_multimae_base(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], **kwargs):
    return MultiMAE(input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=768, depth=12, num_heads=12,
                    mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


@register_model
def pretrain_multimae_large(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], **kwargs):
    return MultiMAE(input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=1024, depth=24, num_heads=16,
                    mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


@register_model
def multivit_base(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], **kwargs):
    return MultiViT(input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=768, depth=12, num_heads=12,
                    mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


@register_model
def multivit_large(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], **kwargs):
    return MultiViT(input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=1024, depth=24, num_heads=16,
                    mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
