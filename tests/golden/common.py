"""Seeded weights / inputs shared by the fixture generator and the tests that replay the
fixtures.  Test infrastructure: imports the oracle, never imported by the product."""
import numpy as np
import torch

from oracle import multimae_oracle as O

SCHEDULE_CASES = [
    dict(base_value=1e-4, final_value=0.0, epochs=20, niter_per_ep=30, warmup_epochs=2, warmup_steps=-1),
    dict(base_value=2.5e-3, final_value=1e-6, epochs=7, niter_per_ep=13, warmup_epochs=1, start_warmup_value=1e-6, warmup_steps=5),
    dict(base_value=0.05, final_value=0.05, epochs=3, niter_per_ep=11),
]

SEMSEG_GRAD_NAMES = [
    "global_tokens",
    "input_adapters.semseg.class_emb.weight",
    "input_adapters.semseg.proj.weight",
    "input_adapters.semseg.proj.bias",
    "input_adapters.rgb.proj.weight",
    "encoder.0.attn.qkv.weight",
    "output_adapters.semseg.out_proj.weight",
    "output_adapters.semseg.out_proj.bias",
    "output_adapters.semseg.task_embeddings.rgb",
    "output_adapters.depth.task_embeddings.semseg",
    "output_adapters.semseg.mask_token",
]

TINY_GRAD_NAMES = [
    "global_tokens",
    "input_adapters.depth.proj.weight",
    "input_adapters.rgb.proj.bias",
    "encoder.0.attn.qkv.weight",
    "encoder.1.mlp.fc2.bias",
    "encoder.1.norm2.weight",
    "output_adapters.rgb.mask_token",
    "output_adapters.rgb.task_embeddings.depth",
    "output_adapters.depth.out_proj.bias",
    "output_adapters.norm_rgb.decoder.kv.weight",
    "output_adapters.depth.proj_context.weight",
]


def _perturb(sd, seed, scale=0.05):
    """Zero/one-initialised tensors (biases, LayerNorm, mask_token) get noise so every
    term of the arithmetic is exercised."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if k.endswith("pos_emb"):
            out[k] = v.clone()
        elif v.ndim <= 1 or "mask_token" in k or "task_embeddings" in k or "global_tokens" in k:
            out[k] = v + scale * torch.randn(v.shape, generator=g)
        else:
            out[k] = v.clone()
    return out


def perturbed_state(cfg, seed):
    return _perturb(O.init_state(cfg, seed), seed + 1000)


def perturbed_block_state(prefix, dim, hidden, seed):
    cfg = O.make_cfg({"rgb": 3}, {}, dim=dim, depth=1, heads=4, mlp_ratio=hidden / dim, image=16)
    sd = O.init_state(cfg, seed)
    sd = {k.replace("encoder.0", prefix): v for k, v in sd.items() if k.startswith("encoder.0.")}
    return _perturb(sd, seed + 1000)


def perturbed_xattn_state(prefix, dim, seed):
    cfg = O.make_cfg({"rgb": 3}, {"rgb": (3, "rgb")}, dim=64, depth=0, dec_dim=dim, dec_depth=0, dec_heads=4, image=16)
    sd = O.init_state(cfg, seed)
    pre = "output_adapters.rgb.decoder."
    sd = {prefix + "." + k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    return _perturb(sd, seed + 1000)


OPTION_GRAD_NAMES = [
    "global_tokens",
    "input_adapters.rgb.proj.bias",
    "input_adapters.rgb.pos_emb",
    "input_adapters.depth.pos_emb",
    "encoder.0.attn.qkv.bias",
    "encoder.0.attn.proj.bias",
    "encoder.1.mlp.fc2.bias",
    "encoder.1.norm1.weight",
    "output_adapters.rgb.mask_token",
    "output_adapters.depth.mask_token",
    "output_adapters.rgb.task_embeddings.rgb",
    "output_adapters.depth.task_embeddings.rgb",
    "output_adapters.depth.task_embeddings.depth",
    "output_adapters.rgb.proj_context.weight",
    "output_adapters.depth.proj_context.bias",
    "output_adapters.rgb.decoder.q.weight",
    "output_adapters.depth.query_norm.weight",
    "output_adapters.depth.mlp.fc1.weight",
    "output_adapters.rgb.decoder_transformer.0.attn.proj.weight",
    "output_adapters.depth.decoder_transformer.1.mlp.fc1.bias",
    "output_adapters.rgb.out_proj.bias",  # (the L1 loss's bias gradient is a sum of signs: ill-conditioned under bf16 predictions)
]


OPTION_BATCH = 4


def option_cases():
    """The module settings outside the pretraining configuration, as variations of small_cfg: name -> (cfg, input-adapter
    keyword arguments per domain, output-adapter keyword arguments, model keyword arguments, training mode)."""
    two = {"rgb": (3, "rgb"), "depth": (1, "depth")}
    base = dict(dim=128, depth=2, heads=4, dec_dim=64, dec_heads=2, patch=16, image=64)
    rgbd = O.make_cfg({"rgb": 3, "depth": 1}, two, dec_depth=1, **base)
    return {
        "noxattn": (rgbd, {}, dict(use_xattn=False), {}, False),
        "ownq": (rgbd, {}, dict(use_task_queries=False), {}, False),
        "rgbonly": (O.make_cfg({"rgb": 3}, two, dec_depth=1, **base), {}, {}, {}, False),
        "learnpos": (rgbd, {"rgb": dict(sincos_pos_emb=False), "depth": dict(learnable_pos_emb=True)}, {}, {}, False),
        "droppath": (O.make_cfg({"rgb": 3, "depth": 1}, two, dec_depth=2, **base), {}, dict(drop_path_rate=0.3),
                     dict(drop_path_rate=0.3), True),
    }


def option_state(cfg, seed, keys, random_tables=()):
    """perturbed_state restricted to `keys` (the state_dict of the variant being built: a KeyError names a key the full
    model does not have); the tables in `random_tables` are replaced by seeded N(0, 0.02) values."""
    sd = perturbed_state(cfg, seed)
    out = {k: sd[k] for k in keys}
    g = torch.Generator().manual_seed(seed + 2000)
    for k in random_tables:
        out[k] = 0.02 * torch.randn(out[k].shape, generator=g)
    return out


def seeded_inputs(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    S = cfg["image"]
    x = {}
    for d, C in cfg["in_domains"].items():
        if d in cfg.get("num_classes", {}):
            st = cfg["stride"][d]
            x[d] = torch.randint(0, cfg["num_classes"][d], (B, S // st, S // st), generator=g)
        elif d == "rgb":
            x[d] = torch.randn(B, C, S, S, generator=g)
        else:
            x[d] = torch.rand(B, C, S, S, generator=g)
    return x


def mask_keys(p, noises, noise_all, k):
    """The fp32 sort keys of the second argsort (preliminary mask + global noise)."""
    kt = O.samples_per_task(p, k)
    pre = [torch.where(torch.argsort(n, dim=1, stable=True) < kt[:, i:i + 1], 0, 1) for i, n in enumerate(noises)]
    return torch.cat(pre, 1) + noise_all


def assert_tie_free(p, noises, noise_all, k):
    for n in noises:
        for row in n:
            assert row.unique().numel() == row.numel(), "tie in per-task noise"
    for row in mask_keys(p, noises, noise_all, k):
        assert row.unique().numel() == row.numel(), "tie in global sort key"


def load(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), name))
