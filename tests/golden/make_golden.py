"""Generate the golden fixtures in this directory by running the REFERENCE implementation.

Run in the build container only (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Every fixture is data: seeds / inputs and the outputs the reference's ``multimae``
package produced for them.  Weights and image inputs are regenerated from seeds by
``oracle.multimae_oracle`` helpers (``perturbed_state``/``seeded_inputs``) and loaded
into the reference modules with ``load_state_dict(strict=True)``, which also pins the
state_dict key names and shapes.
"""
import hashlib
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

from torch.distributions.dirichlet import Dirichlet  # noqa: E402

import multimae as ref  # noqa: E402  (the reference package)
from multimae.multimae_utils import Block as RefBlock, CrossAttention as RefXAttn, build_2d_sincos_posemb  # noqa: E402
from oracle import multimae_oracle as O  # noqa: E402
from tests.golden import common as G  # noqa: E402


def build_ref_model(cfg):
    n_img = cfg["image"]
    ia = {d: ref.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"],
                                     image_size=n_img) for d, C in cfg["in_domains"].items()}
    oa = {name: ref.SpatialOutputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"],
                                         dim_tokens=cfg["dec_dim"], depth=cfg["dec_depth"],
                                         num_heads=cfg["dec_heads"], use_task_queries=True, task=task,
                                         context_tasks=list(cfg["in_domains"]), use_xattn=True,
                                         image_size=n_img)
          for name, (C, task) in cfg["out_domains"].items()}
    return ref.MultiMAE(ia, oa, num_global_tokens=cfg["num_global"], dim_tokens=cfg["dim"],
                        depth=cfg["depth"], num_heads=cfg["heads"])


class _Tokens(dict):
    pass


def replay_mask_rng(seed, B, sizes, alphas, uniform):
    """Draw, in the reference's call order (multimae/multimae.py:211-237), the values its
    generate_random_masks will consume after torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    T = len(sizes)
    al = [alphas] * T if isinstance(alphas, float) else alphas
    if uniform:
        dummy = ref.MultiMAE.sample_alphas(None, B, T, alphas=al)
        p = Dirichlet(dummy).sample()
    else:
        p = Dirichlet(torch.Tensor(al)).sample((B,))
    noises = [torch.rand(B, n) for n in sizes]
    noise_all = torch.rand(B, sum(sizes))
    return p, noises, noise_all


def gen_masks():
    out = {}
    model = build_ref_model(O.tiny_cfg())
    case = 0
    for T in (2, 3):
        for k in (98, 196):
            for seed in range(4):
                B, sizes = 8, [196] * T
                p, noises, noise_all = replay_mask_rng(seed, B, sizes, 1.0, False)
                toks = {f"t{i}": torch.zeros(B, n, 1) for i, n in enumerate(sizes)}
                torch.manual_seed(seed)
                masks, ids_keep, ids_restore = model.generate_random_masks(toks, k, alphas=1.0)
                key = torch.cat([m for m in masks.values()], 1)  # final masks; tie check on real keys below
                G.assert_tie_free(p, noises, noise_all, k)
                pre = f"c{case}_"
                out[pre + "k"] = np.int64(k)
                out[pre + "p"] = p.numpy()
                for i, nz in enumerate(noises):
                    out[pre + f"noise{i}"] = nz.numpy()
                out[pre + "noise_all"] = noise_all.numpy()
                out[pre + "ids_keep"] = ids_keep.numpy().astype(np.int16)
                out[pre + "ids_restore"] = ids_restore.numpy().astype(np.int16)
                out[pre + "mask"] = key.numpy().astype(np.uint8)
                case += 1
    # sample_tasks_uniformly (alphas rows with eps entries), T=2 and T=3
    for T in (2, 3):
        B, sizes, k, seed = 8, [196] * T, 98, 11 + T
        p, noises, noise_all = replay_mask_rng(seed, B, sizes, 1.0, True)
        toks = {f"t{i}": torch.zeros(B, n, 1) for i, n in enumerate(sizes)}
        torch.manual_seed(seed)
        masks, ids_keep, ids_restore = model.generate_random_masks(toks, k, alphas=1.0, sample_tasks_uniformly=True)
        G.assert_tie_free(p, noises, noise_all, k)
        pre = f"c{case}_"
        out[pre + "k"] = np.int64(k)
        out[pre + "p"] = p.numpy()
        for i, nz in enumerate(noises):
            out[pre + f"noise{i}"] = nz.numpy()
        out[pre + "noise_all"] = noise_all.numpy()
        out[pre + "ids_keep"] = ids_keep.numpy().astype(np.int16)
        out[pre + "ids_restore"] = ids_restore.numpy().astype(np.int16)
        out[pre + "mask"] = torch.cat(list(masks.values()), 1).numpy().astype(np.uint8)
        case += 1
    out["n_cases"] = np.int64(case)
    # how often does sum(round(p*k)) differ from k in these cases (must be exercised)
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **out)
    print("masks:", case, "cases")


def gen_posemb():
    out = {}
    for (h, w, d) in [(14, 14, 768), (14, 14, 256), (3, 5, 8), (24, 24, 1024)]:
        t = build_2d_sincos_posemb(h, w, d).contiguous()
        tag = f"{h}x{w}x{d}"
        out[tag + "_sha256"] = np.frombuffer(hashlib.sha256(t.numpy().tobytes()).digest(), dtype=np.uint8)
        flat = t.flatten()
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        out[tag + "_probe_idx"] = idx.numpy()
        out[tag + "_probe_val"] = flat[idx].numpy()
        if flat.numel() <= 4096:
            out[tag + "_full"] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "posemb.npz"), **out)
    print("posemb ok")


def gen_tiny_e2e(cfg=None, fname="tiny_e2e.npz"):
    cfg = cfg or O.tiny_cfg()
    sd = G.perturbed_state(cfg, seed=7)
    model = build_ref_model(cfg)
    missing = model.load_state_dict(sd, strict=True)
    x = G.seeded_inputs(cfg, B=2, seed=3)
    k = 8
    n = (cfg["image"] // cfg["patch"]) ** 2
    seed = 5
    p, noises, noise_all = replay_mask_rng(seed, 2, [n, n], 1.0, False)
    G.assert_tie_free(p, noises, noise_all, k)
    losses_fn = {"rgb": ref.MaskedMSELoss(patch_size=16, stride=1),
                 "depth": ref.MaskedL1Loss(patch_size=16, stride=1),
                 "norm_rgb": ref.MaskedMSELoss(patch_size=16, stride=1, norm_pix=True)}
    tgt = {"rgb": "rgb", "depth": "depth", "norm_rgb": "rgb"}
    out = {"k": np.int64(k), "p": p.numpy(), "noise0": noises[0].numpy(), "noise1": noises[1].numpy(),
           "noise_all": noise_all.numpy()}

    torch.manual_seed(seed)
    model.zero_grad()
    preds, masks = model(x, num_encoded_tokens=k, alphas=1.0)
    losses = {t: losses_fn[t](preds[t].float(), x[tgt[t]], mask=masks[tgt[t]]) for t in preds}
    sum(losses.values()).backward()
    named = dict(model.named_parameters())
    for t in preds:
        out[f"pred_{t}"] = preds[t].detach().numpy()
        out[f"loss_{t}"] = losses[t].detach().numpy()
    for d in masks:
        out[f"mask_{d}"] = masks[d].numpy().astype(np.uint8)
    for name in G.TINY_GRAD_NAMES:
        out["grad_" + name] = named[name].grad.numpy()
    # global gradient norm over all trainable parameters
    out["grad_norm"] = torch.sqrt(sum((p_.grad.double() ** 2).sum() for p_ in named.values() if p_.grad is not None)).numpy()
    out["n_state_keys"] = np.int64(len(model.state_dict()))

    # the same step under bf16 autocast (the reference's AMP semantics with bf16 as the dtype)
    torch.manual_seed(seed)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        preds_b, masks_b = model(x, num_encoded_tokens=k, alphas=1.0)
    for t in preds_b:
        out[f"bf16_loss_{t}"] = losses_fn[t](preds_b[t].float(), x[tgt[t]], mask=masks_b[tgt[t]]).detach().numpy()
    out["bf16_pred_depth"] = preds_b["depth"].float().detach().numpy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "ok; losses", {t: float(v) for t, v in losses.items()})


class _ClassMapAdapter(torch.nn.Module):
    """Stand-in for the input adapter the reference fork lacks (its call site: run_pretraining_multimae.py:72-85):
    class map -> Embedding -> Conv2d(k=s=P) -> + pos_emb, with the parameter names class_emb / proj / pos_emb.
    Test code written for this fixture; everything downstream of it in the fixture is the reference's own code."""

    def __init__(self, num_classes, dim_class_emb, stride_level, patch_size_full, image_size):
        super().__init__()
        self.num_classes, self.dim_class_emb, self.stride_level = num_classes, dim_class_emb, stride_level
        self.P = max(1, patch_size_full // stride_level)
        self.image_size = image_size
        self.dim_tokens = None

    def init(self, dim_tokens=768):
        self.dim_tokens = dim_tokens
        h = self.image_size // (self.stride_level * self.P)
        self.pos_emb = torch.nn.Parameter(build_2d_sincos_posemb(h, h, dim_tokens), requires_grad=False)
        self.class_emb = torch.nn.Embedding(self.num_classes, self.dim_class_emb)
        self.proj = torch.nn.Conv2d(self.dim_class_emb, dim_tokens, kernel_size=self.P, stride=self.P)

    def no_weight_decay(self):
        return {"pos_emb"}

    def forward(self, x):
        e = self.class_emb(x).permute(0, 3, 1, 2)
        t = self.proj(e).flatten(2).transpose(1, 2)
        pos = torch.nn.functional.interpolate(self.pos_emb, size=(x.shape[1] // self.P, x.shape[2] // self.P), mode="bilinear")
        return t + pos.flatten(2).transpose(1, 2)


def gen_semseg_e2e(fname="small_semseg_e2e.npz"):
    """rgb + depth + class map (stride 4) through the reference's MultiMAE / SpatialOutputAdapter(stride_level=4) /
    MaskedCrossEntropyLoss(16, 4); pins the three-domain masking, the stride-4 decoder output and the masked CE."""
    cfg = O.small_semseg_cfg()
    sd = G.perturbed_state(cfg, seed=9)
    n_img = cfg["image"]
    ia = {}
    for d, C in cfg["in_domains"].items():
        if d in cfg["num_classes"]:
            ia[d] = _ClassMapAdapter(cfg["num_classes"][d], C, cfg["stride"][d], cfg["patch"], n_img)
        else:
            ia[d] = ref.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"], image_size=n_img)
    oa = {name: ref.SpatialOutputAdapter(num_channels=C, stride_level=cfg["stride"].get(name, 1), patch_size_full=cfg["patch"],
                                         dim_tokens=cfg["dec_dim"], depth=cfg["dec_depth"], num_heads=cfg["dec_heads"],
                                         use_task_queries=True, task=task, context_tasks=list(cfg["in_domains"]),
                                         use_xattn=True, image_size=n_img)
          for name, (C, task) in cfg["out_domains"].items()}
    model = ref.MultiMAE(ia, oa, num_global_tokens=cfg["num_global"], dim_tokens=cfg["dim"], depth=cfg["depth"],
                         num_heads=cfg["heads"])
    model.load_state_dict(sd, strict=True)
    x = G.seeded_inputs(cfg, B=2, seed=4)
    k, seed = 12, 6
    n = (cfg["image"] // cfg["patch"]) ** 2
    p, noises, noise_all = replay_mask_rng(seed, 2, [n, n, n], 1.0, False)
    G.assert_tie_free(p, noises, noise_all, k)
    losses_fn = {"rgb": ref.MaskedMSELoss(patch_size=16, stride=1), "depth": ref.MaskedL1Loss(patch_size=16, stride=1),
                 "semseg": ref.MaskedCrossEntropyLoss(patch_size=16, stride=4)}
    out = {"k": np.int64(k), "p": p.numpy(), "noise_all": noise_all.numpy()}
    for i, nz in enumerate(noises):
        out[f"noise{i}"] = nz.numpy()
    torch.manual_seed(seed)
    model.zero_grad()
    preds, masks = model(x, num_encoded_tokens=k, alphas=1.0)
    losses = {t: losses_fn[t](preds[t].float(), x[t], mask=masks[t]) for t in preds}
    sum(losses.values()).backward()
    named = dict(model.named_parameters())
    for t in preds:
        out[f"pred_{t}"] = preds[t].detach().numpy()
        out[f"loss_{t}"] = losses[t].detach().numpy()
    for d in masks:
        out[f"mask_{d}"] = masks[d].numpy().astype(np.uint8)
    for name in G.SEMSEG_GRAD_NAMES:
        out["grad_" + name] = named[name].grad.numpy()
    out["grad_norm"] = torch.sqrt(sum((p_.grad.double() ** 2).sum() for p_ in named.values() if p_.grad is not None)).numpy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "ok; losses", {t: float(v) for t, v in losses.items()})


def gen_multivit(fname="small_multivit.npz"):
    """Reference MultiViT (no masking, all 2*16 tokens + global token through the encoder) on the small configuration."""
    cfg = O.small_cfg()
    sd = {k: v for k, v in G.perturbed_state(cfg, seed=7).items() if not k.startswith("output_adapters.")}
    ia = {d: ref.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"], image_size=cfg["image"])
          for d, C in cfg["in_domains"].items()}
    model = ref.MultiViT(ia, None, num_global_tokens=cfg["num_global"], dim_tokens=cfg["dim"], depth=cfg["depth"],
                         num_heads=cfg["heads"])
    model.load_state_dict(sd, strict=True)
    x = G.seeded_inputs(cfg, B=2, seed=3)
    layers = model(x, return_all_layers=True)
    last = model(x)
    assert torch.equal(last, layers[-1])
    g = torch.Generator().manual_seed(77)
    dy = torch.randn(last.shape, generator=g)
    model.zero_grad()
    last.backward(dy)
    named = dict(model.named_parameters())
    out = {"dy": dy.numpy(), "rgb_only": model({"rgb": x["rgb"]}).detach().numpy()}
    for i, t in enumerate(layers):
        out[f"layer{i}"] = t.detach().numpy()
    for n in ("global_tokens", "input_adapters.depth.proj.weight", "encoder.0.attn.qkv.weight", "encoder.1.mlp.fc2.bias"):
        out["grad_" + n] = named[n].grad.numpy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "ok", tuple(last.shape))


def gen_options(fname="options.npz"):
    """The reference's module settings outside the pretraining configuration (no cross-attention, own queries, a decoder
    whose task is not an input, learnable input position tables, stochastic depth in training mode): one step each."""
    out = {}
    for case, (cfg, ia_kw, oa_kw, m_kw, training) in G.option_cases().items():
        S = cfg["image"]
        ia = {d: ref.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"], image_size=S,
                                         **ia_kw.get(d, {})) for d, C in cfg["in_domains"].items()}
        oa = {name: ref.SpatialOutputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"],
                                             dim_tokens=cfg["dec_dim"], depth=cfg["dec_depth"], num_heads=cfg["dec_heads"],
                                             task=task, context_tasks=list(cfg["in_domains"]), image_size=S, **oa_kw)
              for name, (C, task) in cfg["out_domains"].items()}
        model = ref.MultiMAE(ia, oa, num_global_tokens=cfg["num_global"], dim_tokens=cfg["dim"], depth=cfg["depth"],
                             num_heads=cfg["heads"], **m_kw)
        tables = [f"input_adapters.{d}.pos_emb" for d, kw in ia_kw.items() if kw.get("sincos_pos_emb") is False]
        model.load_state_dict(G.option_state(cfg, 7, list(model.state_dict().keys()), tables), strict=True)
        model.train(training)
        B, k, seed = G.OPTION_BATCH, 8, 5
        x_all = G.seeded_inputs(O.small_cfg(), B=B, seed=3)
        x = {d: x_all[d] for d in cfg["in_domains"]}
        n = (S // cfg["patch"]) ** 2
        p, noises, noise_all = replay_mask_rng(seed, B, [n] * len(x), 1.0, False)
        G.assert_tie_free(p, noises, noise_all, k)
        losses_fn = {"rgb": ref.MaskedMSELoss(patch_size=16, stride=1), "depth": ref.MaskedMSELoss(patch_size=16, stride=1)}  # (L1's sign gradients are ill-conditioned under bf16; tested elsewhere)
        pre = case + "_"
        out[pre + "p"], out[pre + "noise_all"] = p.numpy(), noise_all.numpy()
        for i, nz in enumerate(noises):
            out[pre + f"noise{i}"] = nz.numpy()
        # stochastic depth draws (torch.rand of shape [B, 1, 1]) in call order
        draws, real_rand = [], torch.rand

        def logging_rand(*a, **kw):
            r = real_rand(*a, **kw)
            if tuple(r.shape) == (B, 1, 1):
                draws.append(r.clone())
            return r
        torch.manual_seed(seed)
        model.zero_grad()
        torch.rand = logging_rand
        try:
            preds, masks = model(x, num_encoded_tokens=k, alphas=1.0)
        finally:
            torch.rand = real_rand
        # targets of a domain that is not an input come from the full seeded set; its loss then runs without a mask
        losses = {t: losses_fn[t](preds[t].float(), x_all[t], mask=masks.get(t)) for t in preds}
        sum(losses.values()).backward()
        named = dict(model.named_parameters())
        for t in preds:
            out[pre + f"pred_{t}"] = preds[t].detach().numpy().astype(np.float16)  # compared at 2e-2: half precision is ample
            out[pre + f"loss_{t}"] = losses[t].detach().numpy()
        for d in masks:
            out[pre + f"mask_{d}"] = masks[d].numpy().astype(np.uint8)
        for name in G.OPTION_GRAD_NAMES:
            if name in named and named[name].grad is not None:
                out[pre + "grad_" + name] = named[name].grad.numpy()
        out[pre + "grad_norm"] = torch.sqrt(sum((q.grad.double() ** 2).sum() for q in named.values() if q.grad is not None)).numpy()
        out[pre + "n_trainable"] = np.int64(sum(q.numel() for q in named.values() if q.requires_grad))
        if draws:
            out[pre + "droppath_draws"] = torch.stack(draws).numpy()
        print(case, "ok; losses", {t: float(v) for t, v in losses.items()}, "draws", len(draws))
    np.savez_compressed(os.path.join(HERE, fname), **out)


def gen_keys():
    """Digest of the reference's downstream-export key list (utils/multimae_keys.py:4-156) and its length."""
    from utils.multimae_keys import MULTIVIT_PRETRAINED_KEYS
    digest = hashlib.sha256("\n".join(MULTIVIT_PRETRAINED_KEYS).encode()).digest()
    np.savez_compressed(os.path.join(HERE, "keys.npz"), multivit_sha256=np.frombuffer(digest, dtype=np.uint8),
                        multivit_count=np.int64(len(MULTIVIT_PRETRAINED_KEYS)))
    print("keys ok", len(MULTIVIT_PRETRAINED_KEYS))


def gen_schedule():
    import contextlib
    import io
    from utils.native_scaler import cosine_scheduler
    out = {}
    for i, kw in enumerate(G.SCHEDULE_CASES):
        with contextlib.redirect_stdout(io.StringIO()):
            out[f"case{i}"] = cosine_scheduler(**kw)
    np.savez_compressed(os.path.join(HERE, "schedule.npz"), **out)
    print("schedule ok")


def gen_losses():
    out = {}
    g = torch.Generator().manual_seed(21)

    def run(tag, fn, pred, target, mask):
        pred = pred.clone().requires_grad_(True)
        loss = fn(pred, target, mask=mask)
        out[tag + "_loss"] = loss.detach().numpy()
        if loss.requires_grad:
            loss.backward()
            out[tag + "_dpred"] = pred.grad.numpy()
        out[tag + "_pred"] = pred.detach().numpy()
        out[tag + "_target"] = target.numpy()
        if mask is not None:
            out[tag + "_mask"] = mask.numpy().astype(np.uint8)

    B, H = 2, 32  # patch 16 -> 2x2 patches
    for C, tagc in ((3, "c3"), (1, "c1")):
        pred = torch.randn(B, C, H, H, generator=g)
        target = torch.randn(B, C, H, H, generator=g)
        mask = torch.tensor([[1, 0, 1, 1], [0, 0, 1, 0]])
        run(f"mse_{tagc}", ref.MaskedMSELoss(16, 1), pred, target, mask)
        run(f"l1_{tagc}", ref.MaskedL1Loss(16, 1), pred, target, mask)
        run(f"msenp_{tagc}", ref.MaskedMSELoss(16, 1, norm_pix=True), pred, target, mask)
        run(f"l1np_{tagc}", ref.MaskedL1Loss(16, 1, norm_pix=True), pred, target, mask)
        # a sample with no masked token: 0/0 for that sample, dropped by nanmean
        mask0 = torch.tensor([[0, 0, 0, 0], [0, 1, 1, 0]])
        run(f"mse_rowzero_{tagc}", ref.MaskedMSELoss(16, 1), pred, target, mask0)
        # nothing masked at all: int64 zero
        run(f"mse_allzero_{tagc}", ref.MaskedMSELoss(16, 1), pred, target, torch.zeros(B, 4, dtype=torch.long))
        run(f"mse_nomask_{tagc}", ref.MaskedMSELoss(16, 1), pred, target, None)
        run(f"l1_nomask_{tagc}", ref.MaskedL1Loss(16, 1), pred, target, None)
    # cross entropy: 5 classes, stride 4 -> scale 4, 16x16 map = 4x4 patches
    logits = torch.randn(B, 5, 16, 16, generator=g)
    tgt = torch.randint(0, 5, (B, 16, 16), generator=g)
    mask = (torch.rand(B, 16, generator=g) > 0.4).long()
    run("ce", ref.MaskedCrossEntropyLoss(16, 4), logits, tgt, mask)
    run("ce_ls", ref.MaskedCrossEntropyLoss(16, 4, label_smoothing=0.1), logits, tgt, mask)
    run("ce_nomask", ref.MaskedCrossEntropyLoss(16, 4), logits, tgt, None)
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses ok")


def gen_blocks():
    out = {}
    from functools import partial
    g = torch.Generator().manual_seed(33)
    # Block(64, 4 heads), N=99, B=2
    sd = G.perturbed_block_state("blk", 64, 256, seed=41)
    blk = RefBlock(dim=64, num_heads=4, mlp_ratio=4.0, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
    blk.load_state_dict({k[len("blk."):]: v for k, v in sd.items()}, strict=True)
    x = torch.randn(2, 99, 64, generator=g).requires_grad_(True)
    dy = torch.randn(2, 99, 64, generator=g)
    y = blk(x)
    y.backward(dy)
    out["block_x"] = x.detach().numpy(); out["block_dy"] = dy.numpy()
    out["block_y"] = y.detach().numpy(); out["block_dx"] = x.grad.numpy()
    out["block_dqkv_w"] = blk.attn.qkv.weight.grad.numpy()
    out["block_dfc1_b"] = blk.mlp.fc1.bias.grad.numpy()
    out["block_dnorm1_w"] = blk.norm1.weight.grad.numpy()
    # CrossAttention(32, 4 heads), Nq=196, Nctx=99
    sdx = G.perturbed_xattn_state("xa", 32, seed=43)
    xa = RefXAttn(dim=32, num_heads=4, qkv_bias=True)
    xa.load_state_dict({k[len("xa."):]: v for k, v in sdx.items()}, strict=True)
    q = torch.randn(2, 196, 32, generator=g).requires_grad_(True)
    c = torch.randn(2, 99, 32, generator=g).requires_grad_(True)
    dy = torch.randn(2, 196, 32, generator=g)
    y = xa(q, c)
    y.backward(dy)
    out["xa_q"] = q.detach().numpy(); out["xa_c"] = c.detach().numpy(); out["xa_dy"] = dy.numpy()
    out["xa_y"] = y.detach().numpy(); out["xa_dq"] = q.grad.numpy(); out["xa_dc"] = c.grad.numpy()
    out["xa_dkv_w"] = xa.kv.weight.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **out)
    print("blocks ok")


def gen_depth_std():
    # The reference applies this inline in its training loop (run_pretraining_multimae.py:448-459),
    # which is not importable; the three statements are replayed here with the same torch/einops calls.
    from einops import rearrange
    g = torch.Generator().manual_seed(55)
    depth = torch.rand(2, 1, 32, 32, generator=g) * 7.0
    trunc = torch.sort(rearrange(depth, "b c h w -> b (c h w)"), dim=1)[0]
    trunc = trunc[:, int(0.1 * trunc.shape[1]): int(0.9 * trunc.shape[1])]
    res = (depth - trunc.mean(dim=1)[:, None, None, None]) / torch.sqrt(trunc.var(dim=1)[:, None, None, None] + 1e-6)
    np.savez_compressed(os.path.join(HERE, "depth_std.npz"), depth=depth.numpy(), out=res.numpy())
    print("depth std ok")


if __name__ == "__main__":
    if len(sys.argv) > 1:  # regenerate only the named fixtures, e.g. `make_golden.py gen_semseg_e2e`
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    gen_semseg_e2e()
    gen_schedule()
    gen_multivit()
    gen_options()
    gen_keys()
    gen_masks()
    gen_posemb()
    gen_tiny_e2e()
    gen_tiny_e2e(O.small_cfg(), "small_e2e.npz")
    gen_losses()
    gen_blocks()
    gen_depth_std()
