"""End-to-end parity of the HIP MultiMAE step (module API -> C ABI kernels) against the reference's golden
outputs and against the oracle.  bf16 tolerance: 2e-2 relative (L2 over a tensor); masks bit-exact."""
import numpy as np
import pytest
import torch

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
BF16_TOL = 2e-2
FP32_TOL = 1e-4


def _build(cfg):
    import multimae_amd as M
    n_img = cfg["image"]
    ia = {d: M.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"], image_size=n_img)
          for d, C in cfg["in_domains"].items()}
    oa = {name: M.SpatialOutputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"], dim_tokens=cfg["dec_dim"],
                                       depth=cfg["dec_depth"], num_heads=cfg["dec_heads"], use_task_queries=True, task=task,
                                       context_tasks=list(cfg["in_domains"]), use_xattn=True, image_size=n_img)
          for name, (C, task) in cfg["out_domains"].items()}
    return M.MultiMAE(ia, oa, num_global_tokens=cfg["num_global"], dim_tokens=cfg["dim"], depth=cfg["depth"], num_heads=cfg["heads"])


def _losses(cfg):
    import multimae_amd as M
    return {"rgb": (M.MaskedMSELoss(patch_size=cfg["patch"], stride=1), "rgb"),
            "depth": (M.MaskedL1Loss(patch_size=cfg["patch"], stride=1), "depth"),
            "norm_rgb": (M.MaskedMSELoss(patch_size=cfg["patch"], stride=1, norm_pix=True), "rgb")}


def _hip_step(model, cfg, x, rnd, k):
    model.zero_grad(set_to_none=True)
    xg = {d: t.cuda() for d, t in x.items()}
    preds, masks = model(xg, num_encoded_tokens=k, alphas=1.0, mask_randomness=rnd)
    losses = {t: fn(preds[t], xg[tg], mask=masks[tg]) for t, (fn, tg) in _losses(cfg).items()}
    sum(losses.values()).backward()
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    return preds, masks, losses, grads


def test_small_model_vs_reference_golden():
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    z = G.load("small_e2e.npz")
    cfg = O.small_cfg()
    model = _build(cfg)
    model.load_state_dict(G.perturbed_state(cfg, seed=7), strict=True)
    model.cuda()
    x = G.seeded_inputs(cfg, B=2, seed=3)
    rnd = (torch.from_numpy(z["p"]), [torch.from_numpy(z["noise0"]), torch.from_numpy(z["noise1"])], torch.from_numpy(z["noise_all"]))
    preds, masks, losses, grads = _hip_step(model, cfg, x, rnd, int(z["k"]))
    for d in masks:
        assert masks[d].dtype == torch.int64
        assert np.array_equal(masks[d].cpu().numpy(), z[f"mask_{d}"].astype(np.int64)), d
    report = {}
    for t in preds:
        assert preds[t].shape == z[f"pred_{t}"].shape
        report[f"pred_{t}"] = rel_err(preds[t].float().cpu(), z[f"pred_{t}"])
        report[f"loss_{t}"] = abs(losses[t].item() - float(z[f"loss_{t}"])) / abs(float(z[f"loss_{t}"]))
    for name in G.TINY_GRAD_NAMES:
        report["grad_" + name] = rel_err(grads[name].cpu(), z["grad_" + name])
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
    report["grad_norm"] = abs(gn - float(z["grad_norm"])) / float(z["grad_norm"])
    print({k_: f"{v:.2e}" for k_, v in report.items()})
    bad = {k_: v for k_, v in report.items() if not v < BF16_TOL}
    assert not bad, bad


def test_vit_b_rgbd_step_vs_oracle():
    """Config-1 shapes (ViT-B/16 RGB-D 224^2, 98 visible tokens, batch 4): HIP step vs the fp32 oracle on the host."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    cfg = O.vit_b_rgbd_cfg()
    sd = G.perturbed_state(cfg, seed=1)
    B, k = 4, 98
    x = G.seeded_inputs(cfg, B=B, seed=2)
    p, noises, noise_all = O.draw_mask_randomness(B, [196, 196], 1.0, torch.Generator().manual_seed(4))
    o_losses, o_preds, o_masks, o_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k)

    model = _build(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    preds, masks, losses, grads = _hip_step(model, cfg, x, (p, noises, noise_all), k)
    for d in masks:
        assert torch.equal(masks[d].cpu(), o_masks[d]), d
    errs = {}
    for t in preds:
        assert preds[t].dtype == torch.bfloat16 and preds[t].shape == o_preds[t].shape
        errs[f"pred_{t}"] = rel_err(preds[t].float().cpu(), o_preds[t])
        errs[f"loss_{t}"] = abs(losses[t].item() - float(o_losses[t])) / abs(float(o_losses[t]))
    gerr = {n: rel_err(g.cpu(), o_grads[n]) for n, g in grads.items()}
    worst = sorted(gerr.items(), key=lambda kv: -kv[1])[:8]
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
    on = torch.sqrt(sum((g.double() ** 2).sum() for g in o_grads.values())).item()
    errs["grad_norm"] = abs(gn - on) / on
    # all gradients as one vector
    num = sum(((grads[n].cpu().double() - o_grads[n].double()) ** 2).sum() for n in grads).sqrt().item()
    errs["grad_all"] = num / on
    print({k_: f"{v:.2e}" for k_, v in errs.items()}, "worst per-tensor grads:", [(n, f"{e:.2e}") for n, e in worst])
    assert set(grads) == set(o_grads), "every trainable parameter receives a gradient"
    bad = {k_: v for k_, v in errs.items() if not v < BF16_TOL}
    assert not bad, bad
    # per-tensor: the bar applies to every parameter gradient
    badg = {n: e for n, e in gerr.items() if not e < BF16_TOL}
    assert not badg, badg
    # What bf16 costs the reference itself: the same oracle under torch.autocast(bfloat16) on the host (the reference's AMP
    # arithmetic with bf16 as the low-precision type) against its own fp32 run.  The kernels keep scores, probabilities and
    # the decoder's residual stream in fp32 where autocast rounds them, so they must not be further from fp32 than that is
    # (scripts/parity_margin.py tabulates both at 2 / 4 / 8 images: worst tensor 1.2e-2 against 1.5e-2 at this size).
    _, _, _, a_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k, autocast_dtype=torch.bfloat16)
    a_worst = max(rel_err(a_grads[n], o_grads[n]) for n in o_grads)
    a_all = sum(((a_grads[n].double() - o_grads[n].double()) ** 2).sum() for n in o_grads).sqrt().item() / on
    print(f"autocast(bf16) reference against fp32: worst tensor {a_worst:.2e}, all gradients {a_all:.2e}")
    assert worst[0][1] <= 1.25 * a_worst and errs["grad_all"] <= 1.25 * a_all, (worst[0], a_worst, errs["grad_all"], a_all)

    # bitwise reproducibility of the whole step (no atomics, fixed reduction orders)
    preds2, _, losses2, grads2 = _hip_step(model, cfg, x, (p, noises, noise_all), k)
    assert all(torch.equal(preds[t], preds2[t]) for t in preds)
    assert all(torch.equal(grads[n], grads2[n]) for n in grads)


def test_module_api_surface():
    import multimae_amd as M
    from oracle import multimae_oracle as O
    cfg = O.small_cfg()
    model = _build(cfg).cuda()
    x = {"rgb": torch.randn(3, 3, 64, 64, device="cuda"), "depth": torch.rand(3, 1, 64, 64, device="cuda")}
    preds, masks = model(x, num_encoded_tokens=8, alphas=1.0)
    assert set(preds) == {"rgb", "depth", "norm_rgb"} and preds["depth"].shape == (3, 1, 64, 64)
    assert all(m.shape == (3, 16) and m.dtype == torch.int64 for m in masks.values())
    assert sum(int(m.eq(0).sum()) for m in masks.values()) == 3 * 8
    # sample_tasks_uniformly path, list alphas, plain-tensor input, encoder-only model
    preds, masks = model(x, num_encoded_tokens=8, alphas=[0.5, 2.0], sample_tasks_uniformly=True)
    assert torch.isfinite(preds["rgb"].float()).all()
    toks = {"rgb": torch.empty(3, 16, 0, device="cuda"), "depth": torch.empty(3, 16, 0, device="cuda")}
    tm, ids_keep, ids_restore = model.generate_random_masks(toks, 8, alphas=1.0)
    assert ids_keep.shape == (3, 8) and ids_restore.shape == (3, 32) and ids_keep.dtype == torch.int64
    # standalone input adapter == oracle adapter (all tokens)
    ad = model.input_adapters["depth"]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = O.input_adapter(x["depth"].cpu(), sd, "input_adapters.depth", 16)
    assert rel_err(ad(x["depth"]).cpu(), ref) < 1e-2
    # given task_masks
    given = {"rgb": torch.zeros(3, 16, dtype=torch.long, device="cuda"), "depth": torch.ones(3, 16, dtype=torch.long, device="cuda")}
    preds, back = model(x, task_masks=given)
    assert back is given and torch.isfinite(preds["depth"].float()).all()
    with pytest.raises(Exception):
        model({"rgb": x["rgb"].cpu(), "depth": x["depth"].cpu()}, num_encoded_tokens=8)  # no CPU fallback


def test_flat_storage_step_matches_autograd_path():
    """PretrainStep (flat parameter storage, in-place gradient accumulation by the kernels, fused AdamW) gives the
    same gradients as plain autograd delivery, and its AdamW update matches torch.optim.AdamW."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from multimae_amd.train import PretrainStep, split_decay
    cfg = O.small_cfg()
    sd = G.perturbed_state(cfg, seed=7)
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    ref = _build(cfg); ref.load_state_dict(sd); ref.cuda()
    _, _, _, g_ref = _hip_step(ref, cfg, {d: t.cpu() for d, t in x.items()}, rnd, 8)
    decay, no_decay = split_decay(ref)
    opt = torch.optim.AdamW([{"params": [p for _, p in decay], "weight_decay": 0.05},
                             {"params": [p for _, p in no_decay], "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95))
    opt.step()

    model = _build(cfg); model.load_state_dict(sd); model.cuda()
    step = PretrainStep(model, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, keep_grads=True)
    loss, task_losses, gn2 = step(x, mask_randomness=rnd)
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    assert all(g.data_ptr() >= step.flat.grads.data_ptr() for g in grads.values())
    worst = max(rel_err(grads[n], g_ref[n]) for n in g_ref)
    assert worst < 1e-5, worst
    gn_ref = sum((g.double() ** 2).sum() for g in g_ref.values()).item()
    assert abs(gn2.item() - gn_ref) / gn_ref < 1e-4
    worst_p = max(rel_err(p.detach(), dict(ref.named_parameters())[n].detach()) for n, p in model.named_parameters() if p.requires_grad)
    assert worst_p < 1e-5, worst_p
    # a second step runs on the updated weights (bf16 weight copies must be refreshed)
    loss2, _, _ = step(x, mask_randomness=rnd)
    assert torch.isfinite(loss2) and loss2.item() != loss.item()


def test_vit_l_384_shapes_vs_oracle():
    """Shapes of the ViT-L/16 384^2 configuration (dim 1024, 16 heads, 576 tokens per domain, 288 visible + 1 global,
    decoders over 576 queries) on a 2-layer encoder / 1-layer decoders so the host oracle stays fast."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    out = {"rgb": (3, "rgb"), "depth": (1, "depth"), "norm_rgb": (3, "rgb")}
    cfg = O.make_cfg({"rgb": 3, "depth": 1}, out, dim=1024, depth=2, heads=16, dec_dim=256, dec_depth=1, dec_heads=8, image=384)
    sd = G.perturbed_state(cfg, seed=11)
    B, k = 2, 288
    x = G.seeded_inputs(cfg, B=B, seed=12)
    p, noises, noise_all = O.draw_mask_randomness(B, [576, 576], 1.0, torch.Generator().manual_seed(13))
    o_losses, o_preds, o_masks, o_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k)
    model = _build(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    preds, masks, losses, grads = _hip_step(model, cfg, x, (p, noises, noise_all), k)
    for d in masks:
        assert torch.equal(masks[d].cpu(), o_masks[d]), d
    errs = {f"pred_{t}": rel_err(preds[t].float().cpu(), o_preds[t]) for t in preds}
    errs.update({f"loss_{t}": abs(losses[t].item() - float(o_losses[t])) / abs(float(o_losses[t])) for t in losses})
    gerr = {n: rel_err(g.cpu(), o_grads[n]) for n, g in grads.items()}
    print({k_: f"{v:.2e}" for k_, v in errs.items()}, "worst grads:", sorted(((e, n) for n, e in gerr.items()), reverse=True)[:4])
    assert all(v < BF16_TOL for v in errs.values()), errs
    assert all(e < BF16_TOL for e in gerr.values()), {n: e for n, e in gerr.items() if e >= BF16_TOL}


def test_flat_bf16_weight_copies_track_the_master():
    """Flat storage keeps bf16 W / W^T views current across optimizer steps and re-syncs them after an
    autograd-visible write (load_state_dict, copy_)."""
    from multimae_amd import functional as Fn
    from multimae_amd.train import FlatAdamW, FlatParams
    from oracle import multimae_oracle as O
    cfg = O.small_cfg()
    model = _build(cfg).cuda()
    flat = FlatParams(model)
    opt = FlatAdamW(flat, lr=1e-2, weight_decay=0.05)
    w = model.encoder[0].mlp.fc1.weight
    wb, wt = Fn.bf16_weights(w)
    assert torch.equal(wb, w.detach().bfloat16()) and torch.equal(wt, w.detach().bfloat16().t())
    flat.grads.normal_()
    opt.step()
    wb, wt = Fn.bf16_weights(w)
    assert torch.equal(wb, w.detach().bfloat16()) and torch.equal(wt, w.detach().bfloat16().t())
    with torch.no_grad():
        w.mul_(1.5)
    wb, wt = Fn.bf16_weights(w)
    assert torch.equal(wb, w.detach().bfloat16()) and torch.equal(wt, w.detach().bfloat16().t())


def _build_semseg(cfg):
    import multimae_amd as M
    n_img = cfg["image"]
    ia = {}
    for d, C in cfg["in_domains"].items():
        if d in cfg["num_classes"]:
            ia[d] = M.SemSegInputAdapter(num_classes=cfg["num_classes"][d], dim_class_emb=C, interpolate_class_emb=False,
                                         stride_level=cfg["stride"][d], patch_size_full=cfg["patch"], image_size=n_img)
        else:
            ia[d] = M.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"], image_size=n_img)
    oa = {name: M.SpatialOutputAdapter(num_channels=C, stride_level=cfg["stride"].get(name, 1), patch_size_full=cfg["patch"],
                                       dim_tokens=cfg["dec_dim"], depth=cfg["dec_depth"], num_heads=cfg["dec_heads"],
                                       use_task_queries=True, task=task, context_tasks=list(cfg["in_domains"]), use_xattn=True,
                                       image_size=n_img)
          for name, (C, task) in cfg["out_domains"].items()}
    return M.MultiMAE(ia, oa, num_global_tokens=cfg["num_global"], dim_tokens=cfg["dim"], depth=cfg["depth"], num_heads=cfg["heads"])


def test_three_domain_semseg_model_vs_reference_golden():
    """rgb + depth + class map at stride 4 with a masked-CE semseg decoder (fp32 logits), against the fixture produced by the
    reference's MultiMAE / SpatialOutputAdapter / MaskedCrossEntropyLoss."""
    import multimae_amd as M
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    z = G.load("small_semseg_e2e.npz")
    cfg = O.small_semseg_cfg()
    model = _build_semseg(cfg)
    model.load_state_dict(G.perturbed_state(cfg, seed=9), strict=True)
    model.cuda()
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=2, seed=4).items()}
    rnd = (torch.from_numpy(z["p"]), [torch.from_numpy(z[f"noise{i}"]) for i in range(3)], torch.from_numpy(z["noise_all"]))
    fns = {"rgb": M.MaskedMSELoss(16, 1), "depth": M.MaskedL1Loss(16, 1), "semseg": M.MaskedCrossEntropyLoss(16, 4)}
    model.zero_grad(set_to_none=True)
    preds, masks = model(x, num_encoded_tokens=int(z["k"]), alphas=1.0, mask_randomness=rnd, fp32_output_adapters=["semseg"])
    assert preds["semseg"].dtype == torch.float32 and preds["rgb"].dtype == torch.bfloat16
    losses = {t: fns[t](preds[t], x[t], mask=masks[t]) for t in preds}
    sum(losses.values()).backward()
    for d in masks:
        assert np.array_equal(masks[d].cpu().numpy(), z[f"mask_{d}"].astype(np.int64)), d
    report = {}
    for t in preds:
        assert preds[t].shape == z[f"pred_{t}"].shape
        report[f"pred_{t}"] = rel_err(preds[t].float().cpu(), z[f"pred_{t}"])
        report[f"loss_{t}"] = abs(losses[t].item() - float(z[f"loss_{t}"])) / abs(float(z[f"loss_{t}"]))
    named = dict(model.named_parameters())
    for name in G.SEMSEG_GRAD_NAMES:
        report["grad_" + name] = rel_err(named[name].grad.cpu(), z["grad_" + name])
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None)).item()
    report["grad_norm"] = abs(gn - float(z["grad_norm"])) / float(z["grad_norm"])
    print(report)
    bad = {k: v for k, v in report.items() if not v < BF16_TOL}
    assert not bad, bad
    assert all(p.grad is not None for p in named.values() if p.requires_grad)


def test_fp32_output_adapter_matches_the_fp32_oracle_on_the_same_encoder_tokens():
    """fp32_output_adapters (multimae/multimae.py:426-436 of the reference: the listed decoders run outside autocast): the
    semseg decoder of the ViT-B configuration (133 classes at stride 4, decoder width 256, depth 2) fed the SAME fp32 encoder
    tokens as the oracle's decoder, so that what is compared is the decoder's own arithmetic: logits, masked cross entropy,
    every decoder parameter gradient and the gradient of the encoder tokens to 1e-4 (L2 norm per tensor)."""
    import multimae_amd as M
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    cfg = O.vit_b_rgbd_semseg_cfg(extra_norm_pix=False)
    sd = G.perturbed_state(cfg, seed=31)
    B, k, D = 3, 98, cfg["dim"]
    g = torch.Generator().manual_seed(32)
    p, noises, noise_all = O.draw_mask_randomness(B, [196] * 3, 1.0, g)
    o_masks, ids_keep, ids_restore = O.generate_masks(p, noises, noise_all, k)
    enc = torch.randn(B, k + cfg["num_global"], D, generator=g) * 1.5
    target = torch.randint(0, 133, (B, 56, 56), generator=g)
    prefix = "output_adapters.semseg"
    # oracle: fp32 decoder + masked CE, gradients of its parameters and of the encoder tokens
    osd = {n: (t.clone().requires_grad_(True) if n.startswith(prefix) and t.is_floating_point() and not n.endswith("pos_emb") else t)
           for n, t in sd.items()}
    enc_o = enc.clone().requires_grad_(True)
    names = list(cfg["in_domains"].keys())
    o_pred = O.output_adapter(enc_o, osd, prefix, cfg, 133, "semseg", ids_keep, ids_restore, (224, 224), patch=O.domain_patch(cfg, "semseg"))
    o_loss = O.masked_ce(o_pred, target, o_masks[names.index("semseg")], patch_size=16, stride=4)
    o_loss.backward()
    # this package: the adapter alone, fp32_output=True
    model = _build_semseg(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    adapter = model.output_adapters["semseg"]
    info = model.generate_input_info({d: torch.empty(B, 196, 0) for d in names}, image_size=(224, 224))
    enc_g = enc.cuda().requires_grad_(True)
    pred = adapter(encoder_tokens=enc_g, input_info=info, ids_keep=ids_keep.cuda(), ids_restore=ids_restore.cuda(), fp32_output=True)
    assert pred.dtype == torch.float32 and pred.shape == (B, 133, 56, 56)
    loss = M.MaskedCrossEntropyLoss(16, 4)(pred, target.cuda(), mask=o_masks[names.index("semseg")].cuda())
    loss.backward()
    errs = {"pred": rel_err(pred.detach().cpu(), o_pred.detach()), "loss": abs(loss.item() - o_loss.item()) / abs(o_loss.item()),
            "d_enc": rel_err(enc_g.grad.cpu(), enc_o.grad)}
    for n, q in adapter.named_parameters():
        if q.requires_grad:
            errs[n] = rel_err(q.grad.cpu(), osd[f"{prefix}.{n}"].grad)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print("fp32 decoder against the fp32 oracle:", [(n, f"{e:.1e}") for n, e in worst])
    assert all(e < FP32_TOL for e in errs.values()), worst
    # the same decoder in the step's usual arithmetic (bf16 operands) for comparison: two orders of magnitude further away
    adapter.zero_grad(set_to_none=True)
    pred_b = adapter(encoder_tokens=enc.cuda(), input_info=info, ids_keep=ids_keep.cuda(), ids_restore=ids_restore.cuda())
    assert pred_b.dtype == torch.bfloat16 and rel_err(pred_b.float().cpu(), o_pred.detach()) > 10 * errs["pred"]


def test_vit_b_rgb_depth_semseg_step_vs_oracle():
    """ViT-B/16 with rgb + depth + 133-class map at stride 4 (588 tokens, 98 visible), decoders rgb / depth / semseg /
    norm_rgb, semseg logits in fp32, batch 4: HIP step vs the fp32 oracle, every parameter gradient, reproducibility."""
    import multimae_amd as M
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    cfg = O.vit_b_rgbd_semseg_cfg(extra_norm_pix=True)
    sd = G.perturbed_state(cfg, seed=21)
    B, k = 4, 98
    x = G.seeded_inputs(cfg, B=B, seed=22)
    p, noises, noise_all = O.draw_mask_randomness(B, [196] * 3, 1.0, torch.Generator().manual_seed(23))
    o_losses, o_preds, o_masks, o_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k)
    model = _build_semseg(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    fns = {"rgb": (M.MaskedMSELoss(16, 1), "rgb"), "depth": (M.MaskedL1Loss(16, 1), "depth"),
           "semseg": (M.MaskedCrossEntropyLoss(16, 4), "semseg"), "norm_rgb": (M.MaskedMSELoss(16, 1, norm_pix=True), "rgb")}
    xg = {d: t.cuda() for d, t in x.items()}

    def step():
        model.zero_grad(set_to_none=True)
        preds, masks = model(xg, num_encoded_tokens=k, alphas=1.0, mask_randomness=(p, noises, noise_all),
                             fp32_output_adapters=["semseg"])
        losses = {t: fn(preds[t], xg[tg], mask=masks[tg]) for t, (fn, tg) in fns.items()}
        sum(losses.values()).backward()
        return preds, masks, losses, {n: q.grad for n, q in model.named_parameters() if q.requires_grad}
    preds, masks, losses, grads = step()
    for d in masks:
        assert torch.equal(masks[d].cpu(), o_masks[d]), d
    assert preds["semseg"].shape == (B, 133, 56, 56) and preds["semseg"].dtype == torch.float32
    errs = {f"pred_{t}": rel_err(preds[t].float().cpu(), o_preds[t]) for t in preds}
    errs.update({f"loss_{t}": abs(losses[t].item() - float(o_losses[t])) / abs(float(o_losses[t])) for t in losses})
    gerr = {n: rel_err(g.cpu(), o_grads[n]) for n, g in grads.items()}
    print({k_: f"{v:.2e}" for k_, v in errs.items()}, "worst grads:", sorted(((e, n) for n, e in gerr.items()), reverse=True)[:4])
    assert set(grads) == set(o_grads)
    assert all(v < BF16_TOL for v in errs.values()), errs
    assert all(e < BF16_TOL for e in gerr.values()), {n: e for n, e in gerr.items() if e >= BF16_TOL}
    preds2, _, _, grads2 = step()
    assert all(torch.equal(preds[t], preds2[t]) for t in preds) and all(torch.equal(grads[n], grads2[n]) for n in grads)


def test_pretrain_step_applies_schedules_and_clipping():
    """PretrainStep with per-iteration lr / weight-decay arrays and gradient clipping equals clip_grad_norm_ + torch AdamW
    with the same per-step values, fed the gradients of the same steps.  (The reference optimizer gets the step's own
    gradients rather than those of a second model stepping beside it: the key bias has no gradient in exact arithmetic,
    what arrives is rounding noise that Adam normalises to steps of +-lr, and two models agree on noise only as long as
    every bit of their weights agrees.  Gradient delivery itself is test_flat_storage_step_matches_autograd_path.)"""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from multimae_amd.train import PretrainStep, cosine_scheduler, split_decay
    cfg = O.small_cfg()
    sd = G.perturbed_state(cfg, seed=7)
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    lrs = cosine_scheduler(1e-3, 1e-5, 1, 4, warmup_epochs=1, warmup_steps=2)
    wds = cosine_scheduler(0.05, 0.1, 1, 4)
    clip = 0.5

    model = _build(cfg); model.load_state_dict(sd); model.cuda()
    decay, no_decay = split_decay(model)
    ref = {n: torch.nn.Parameter(p.detach().clone()) for n, p in decay + no_decay}
    opt = torch.optim.AdamW([{"params": [ref[n] for n, _ in decay], "weight_decay": 0.05},
                             {"params": [ref[n] for n, _ in no_decay], "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95))
    step = PretrainStep(model, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, clip_grad=clip,
                        lr_schedule_values=lrs, wd_schedule_values=wds)
    for it in range(3):
        opt.param_groups[0]["lr"] = opt.param_groups[1]["lr"] = float(lrs[it])
        opt.param_groups[0]["weight_decay"] = float(wds[it])
        loss, _, gn2 = step(x, mask_randomness=rnd)
        for n, p in model.named_parameters():
            if p.requires_grad:
                ref[n].grad = p.grad.detach().clone()  # the step's gradients (clipping is a factor inside the optimizer kernel)
        total = torch.nn.utils.clip_grad_norm_(list(ref.values()), clip)
        opt.step()
        assert abs(gn2.sqrt().item() - total.item()) / total.item() < 1e-4 and total.item() > clip  # clipping is active
        worst = max(rel_err(p.detach(), ref[n].detach()) for n, p in model.named_parameters() if p.requires_grad)
        assert worst < 2e-6, (it, worst)


def test_standalone_mlp_attention_cross_attention_vs_oracle():
    """Mlp, Attention and CrossAttention called directly (reference: multimae/multimae_utils.py:162-169, :191-207,
    :224-247), forward and backward against the oracle's restatement."""
    from multimae_amd.multimae_utils import Attention, CrossAttention, Mlp
    from oracle import multimae_oracle as O
    torch.manual_seed(31)
    B, N, M, D, heads = 3, 50, 21, 128, 4
    x = torch.randn(B, N, D); ctx = torch.randn(B, M, D); dy = torch.randn(B, N, D)

    def run(mod, ref_fn, inputs):
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in mod.state_dict().items()}
        ins = [t.clone().requires_grad_(True) for t in inputs]
        y_ref = ref_fn(sd, *ins)
        y_ref.backward(dy)
        mod = mod.cuda()
        gin = [t.cuda().requires_grad_(True) for t in inputs]
        y = mod(*gin)
        y.backward(dy.cuda())
        errs = {"y": rel_err(y.detach().cpu(), y_ref.detach())}
        for i, (a, b) in enumerate(zip(gin, ins)):
            errs[f"din{i}"] = rel_err(a.grad.cpu(), b.grad)
        for n, p in mod.named_parameters():
            errs[n] = rel_err(p.grad.cpu(), sd[n].grad)
        assert all(v < BF16_TOL for v in errs.values()), errs

    mlp, att, xatt = Mlp(D, 4 * D), Attention(D, heads, qkv_bias=True), CrossAttention(D, heads, qkv_bias=True)
    for m in (mlp, att, xatt):
        for n, p in m.named_parameters():
            torch.nn.init.normal_(p, std=0.08 if p.ndim > 1 else 0.02)
    run(mlp, lambda sd, a: O._mlp(a, {"m." + k: v for k, v in sd.items()}, "m"), [x])
    run(att, lambda sd, a: O.self_attention(a, {"a." + k: v for k, v in sd.items()}, "a", heads), [x])
    run(xatt, lambda sd, a, c: O.cross_attention(a, c, {"x." + k: v for k, v in sd.items()}, "x", heads), [x, ctx])


def test_multivit_encoder_vs_reference_golden():
    import multimae_amd as M
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    z = G.load("small_multivit.npz")
    cfg = O.small_cfg()
    sd = {k: v for k, v in G.perturbed_state(cfg, seed=7).items() if not k.startswith("output_adapters.")}
    ia = {d: M.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"], image_size=cfg["image"])
          for d, C in cfg["in_domains"].items()}
    model = M.MultiViT(ia, None, num_global_tokens=cfg["num_global"], dim_tokens=cfg["dim"], depth=cfg["depth"], num_heads=cfg["heads"])
    model.load_state_dict(sd, strict=True)
    model.cuda()
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=2, seed=3).items()}
    layers = model(x, return_all_layers=True)
    assert len(layers) == cfg["depth"] and layers[-1].shape == (2, 33, cfg["dim"])
    for i, t in enumerate(layers):
        assert rel_err(t.detach().cpu(), z[f"layer{i}"]) < BF16_TOL, i
    model.zero_grad(set_to_none=True)
    last = model(x)
    assert torch.equal(last, layers[-1])
    last.backward(torch.from_numpy(z["dy"]).cuda())
    named = dict(model.named_parameters())
    for n in ("global_tokens", "input_adapters.depth.proj.weight", "encoder.0.attn.qkv.weight", "encoder.1.mlp.fc2.bias"):
        assert rel_err(named[n].grad.cpu(), z["grad_" + n]) < BF16_TOL, n
    assert rel_err(model({"rgb": x["rgb"]}).detach().cpu(), z["rgb_only"]) < BF16_TOL
    assert isinstance(M.create_model("multivit_base", input_adapters=ia, output_adapters=None), M.MultiViT)


def test_multivit_longer_than_one_attention_call_vs_oracle():
    """MultiViT without masking at a size whose token count (2 x 324 + 1 = 649 at head_dim 64) is beyond what one attention
    call keeps in LDS (544): the encoder runs its attention in chunks; outputs and gradients against the oracle."""
    import multimae_amd as M
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    cfg = O.make_cfg({"rgb": 3, "depth": 1}, {}, dim=128, depth=2, heads=2, patch=16, image=288)
    sd = G.perturbed_state(cfg, seed=11)
    ia = {d: M.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=16, image_size=288) for d, C in cfg["in_domains"].items()}
    model = M.MultiViT(ia, None, dim_tokens=128, depth=2, num_heads=2)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    x = G.seeded_inputs(cfg, B=2, seed=4)
    ref_params = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith("pos_emb")) for k, v in sd.items()}
    ref = O.multivit_forward(ref_params, cfg, x)
    assert ref.shape == (2, 649, 128)
    dy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5))
    ref.backward(dy)
    out = model({d: t.cuda() for d, t in x.items()})
    assert rel_err(out.detach().cpu(), ref.detach()) < BF16_TOL
    out.backward(dy.cuda())
    named = dict(model.named_parameters())
    for n in ("global_tokens", "input_adapters.rgb.proj.weight", "encoder.0.attn.qkv.weight", "encoder.0.attn.proj.bias", "encoder.1.norm1.weight"):
        assert rel_err(named[n].grad.cpu(), ref_params[n].grad) < BF16_TOL, n


def test_checkpoint_round_trip_and_optimizer_interchange(tmp_path):
    """save_model / auto_load_model reproduce the training state exactly; the optimizer entry loads into
    torch.optim.AdamW (and back); the downstream export holds the encoder-side tensors only."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from multimae_amd import checkpoint as CK
    from multimae_amd.train import PretrainStep, split_decay
    cfg = O.small_cfg()
    sd = G.perturbed_state(cfg, seed=7)
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))

    def fresh():
        m = _build(cfg); m.load_state_dict(sd); m.cuda()
        return m, PretrainStep(m, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05)
    model, step = fresh()
    for _ in range(2):
        step(x, mask_randomness=rnd)
    path = CK.save_model(str(tmp_path), 4, model, step.opt)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"model", "optimizer", "epoch", "scaler", "args"} and set(ck["model"]) == set(sd)

    model2, step2 = fresh()
    assert CK.auto_load_model(str(tmp_path), model2, step2.opt) == 5
    l1, _, _ = step(x, mask_randomness=rnd)
    l2, _, _ = step2(x, mask_randomness=rnd)
    assert torch.equal(l1, l2)
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), model2.state_dict().values()))

    # the optimizer entry is a torch.optim.AdamW state_dict over [decayed, not decayed]
    model3, _ = fresh()
    model3.load_state_dict(ck["model"])
    decay, no_decay = split_decay(model3)
    topt = torch.optim.AdamW([{"params": [p for _, p in decay], "weight_decay": 0.05},
                              {"params": [p for _, p in no_decay], "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95))
    topt.load_state_dict(ck["optimizer"])
    _hip_step(model3, cfg, {d: t.cpu() for d, t in x.items()}, rnd, 8)
    topt.step()
    named = dict(model.named_parameters())
    worst = max(rel_err(p.detach(), named[n].detach()) for n, p in model3.named_parameters() if p.requires_grad)
    assert worst < 2e-5, worst
    # and back: torch's state into the flat optimizer
    model4, step4 = fresh()
    model4.load_state_dict({k: v.detach().cpu() for k, v in model3.state_dict().items()})
    step4.flat.sync_bf16()
    CK.load_optimizer_state_dict(step4.opt, model4, topt.state_dict())
    assert step4.opt.step_count == 3
    l3, _, _ = step(x, mask_randomness=rnd)
    l4, _, _ = step4(x, mask_randomness=rnd)
    # two parameter sets 2e-5 apart, each through its own bf16 forward: the losses agree to the rounding noise of that
    # forward (a few 1e-4 on this 4-image batch), not to the distance of the parameters
    assert abs(l3.item() - l4.item()) / abs(l3.item()) < 5e-4

    exp = torch.load(CK.save_model_for_downstream_tasks(str(tmp_path), "run", 4, model), map_location="cpu", weights_only=True)
    assert list(exp) == ["model"] and not any(k.startswith("output_adapters.") for k in exp["model"])
    assert set(exp["model"]) == {k for k in sd if not k.startswith("output_adapters.")}


def test_bounded_run_ahead_keeps_results_and_event_count():
    """max_steps_in_flight bounds the host's run-ahead with events of earlier steps: at most that many are pending, and
    the step's results do not depend on the bound (0 = unbounded)."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from multimae_amd.train import PretrainStep
    cfg = O.small_cfg()
    sd = G.perturbed_state(cfg, seed=7)
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    losses = {}
    for bound in (0, 1, 2):
        m = _build(cfg); m.load_state_dict(sd); m.cuda()
        step = PretrainStep(m, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, max_steps_in_flight=bound)
        for _ in range(4):
            loss, _, _ = step(x, mask_randomness=rnd)
            assert len(step._step_events) <= max(bound, 0)
        losses[bound] = loss.item()
    assert losses[0] == losses[1] == losses[2]


def test_reference_style_loop_with_utils_helpers():
    """The reference's loop shape (zero_grad, loss_scaler(loss, optimizer, clip_grad=..., parameters=...), per-iteration
    param_group["lr"]) over create_optimizer / NativeScalerWithGradNormCount equals PretrainStep."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from multimae_amd import utils as U
    from multimae_amd.train import PretrainStep
    cfg = O.small_cfg()
    sd = G.perturbed_state(cfg, seed=7)
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    lrs = U.cosine_scheduler(1e-3, 1e-5, 1, 3)
    a = _build(cfg); a.load_state_dict(sd); a.cuda()
    step = PretrainStep(a, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, clip_grad=0.5, lr_schedule_values=lrs)
    b = _build(cfg); b.load_state_dict(sd); b.cuda()
    optimizer, buckets = U.create_optimizer(b, lr=1e-3, weight_decay=0.05)
    loss_scaler = U.NativeScalerWithGradNormCount(buckets=buckets)
    fns = _losses(cfg)
    for it in range(3):
        la, _, gn2 = step(x, mask_randomness=rnd)
        for group in optimizer.param_groups:
            group["lr"] = lrs[it] * group["lr_scale"]
        preds, masks = b(x, num_encoded_tokens=8, alphas=1.0, mask_randomness=rnd)
        loss = sum(fn(preds[t].float(), x[tg], mask=masks[tg]) for t, (fn, tg) in fns.items())
        optimizer.zero_grad()
        norm = loss_scaler(loss, optimizer, clip_grad=0.5, parameters=b.parameters())
        assert torch.equal(loss.detach(), la) and torch.equal(norm, gn2.sqrt())
    assert all(torch.equal(p, q) for p, q in zip(a.state_dict().values(), b.state_dict().values()))
    assert U.get_rank() == 0 and U.get_world_size() == 1 and U.is_main_process()


def test_decoder_streams_do_not_change_the_step():
    """Decoders dealt over the main stream and a decoder stream (default) against everything on one stream: same kernels
    in the same order per decoder, so three optimisation steps must end on bitwise identical parameters."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from multimae_amd import functional as Fn
    from multimae_amd.train import PretrainStep
    cfg = O.small_cfg()
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    res, saved = [], Fn.DECODER_STREAMS
    try:
        for flag in (True, False):
            Fn.DECODER_STREAMS = flag
            model = _build(cfg)
            model.load_state_dict(G.perturbed_state(cfg, seed=7))
            model.cuda()
            step = PretrainStep(model, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, clip_grad=1.0)
            for _ in range(3):
                loss, _, gn2 = step(x, mask_randomness=rnd)
            torch.cuda.synchronize()
            res.append((step.flat.params.clone(), loss.clone(), gn2.clone()))
    finally:
        Fn.DECODER_STREAMS = saved
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_bucketwise_optimizer_equals_the_update_at_the_end_of_the_step():
    """Without clipping the optimizer runs bucket by bucket beside backward (PretrainStep.eager); the element-wise AdamW
    arithmetic is the same, so three steps must end on bitwise the parameters, moments and bf16 weight copies of the
    update applied once at the end; with clipping the step falls back to the end-of-step update.  The bucket-wise update
    also leaves the squared gradient norm (partial sums per bucket, so equal to the separate pass up to the order of the
    additions) and clears the gradients behind its read unless keep_grads is set."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from multimae_amd import functional as Fn
    from multimae_amd.train import PretrainStep
    cfg = O.small_cfg()
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    res, norms, grads = [], [], []
    for eager, keep in ((True, False), (True, True), (False, False)):
        model = _build(cfg)
        model.load_state_dict(G.perturbed_state(cfg, seed=7))
        model.cuda()
        step = PretrainStep(model, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, bucket_mb=0.25, eager_optimizer=eager,
                            keep_grads=keep)
        assert step.eager == eager and len(step.buckets.buckets) > 3
        for _ in range(3):
            loss, _, gn2 = step(x, mask_randomness=rnd)
        torch.cuda.synchronize()
        wt = torch.cat([Fn.bf16_weights(q)[1].flatten() for q in model.parameters() if q.requires_grad and q.ndim == 2])
        res.append((step.flat.params.clone(), step.opt.exp_avg.clone(), step.opt.exp_avg_sq.clone(), step.flat.w_bf16.clone(),
                    wt, loss.clone()))
        norms.append(gn2.item())
        grads.append(step.flat.grads.clone())
        assert step.opt.step_count == 3
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)
    assert norms[0] == norms[1] and abs(norms[0] - norms[2]) <= 1e-6 * norms[2] and norms[2] > 0
    assert not grads[0].any()                      # cleared by the bucket updates: the next step starts without a fill pass
    assert torch.equal(grads[1], grads[2]) and grads[1].abs().sum() > 0
    assert abs(grads[1].double().pow(2).sum().item() - norms[1]) <= 1e-5 * norms[1]
    clipped = PretrainStep(_build(cfg).cuda(), _losses(cfg), num_encoded_tokens=8, clip_grad=1.0)
    assert not clipped.eager
    # a caller that drops the gradients between steps (model.zero_grad(set_to_none=True)) or runs a backward pass of its own
    # does not disturb the step that skips its fill pass
    model = _build(cfg)
    model.load_state_dict(G.perturbed_state(cfg, seed=7))
    model.cuda()
    step = PretrainStep(model, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, bucket_mb=0.25)
    step(x, mask_randomness=rnd)
    model.zero_grad(set_to_none=True)
    step(x, mask_randomness=rnd)
    preds, masks = model(x, num_encoded_tokens=8, mask_randomness=rnd)
    sum(fn(preds[t], x[tgt], mask=masks[tgt]) for t, (fn, tgt) in _losses(cfg).items()).backward()  # leaves gradients behind
    Fn.join_side_stream()
    assert step.flat.grads.any()
    step(x, mask_randomness=rnd)
    torch.cuda.synchronize()
    assert torch.equal(step.flat.params, res[0][0]) and not step.flat.grads.any()


def test_full_batch_step_is_the_mean_of_its_chunks():
    """The bench size (ViT-B/16 RGB-D, 256 images, 98 visible tokens) through a property that does not need the host
    oracle at that size: every sample is processed independently and each loss is a mean over samples, so predictions of
    the full batch equal those of its 4-image chunks, and losses / gradients equal the mean over the 64 chunks.  The
    chunk size is the one test_vit_b_rgbd_step_vs_oracle pins against the oracle.  Different batch sizes take different
    GEMM kernels and split the weight-gradient reductions differently, so gradients are compared to fp32 rounding (1e-5
    for all of them as one vector, 1e-4 per tensor; measured 1e-7 and 3e-7), not bitwise; the full-size step itself must
    be bitwise reproducible."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    cfg = O.vit_b_rgbd_cfg()
    B, C, k = 256, 4, 98
    model = _build(cfg)
    model.load_state_dict(G.perturbed_state(cfg, seed=1), strict=True)
    model.cuda()
    x = G.seeded_inputs(cfg, B=B, seed=5)
    p, noises, noise_all = O.draw_mask_randomness(B, [196, 196], 1.0, torch.Generator().manual_seed(6))
    preds, masks, losses, grads = _hip_step(model, cfg, x, (p, noises, noise_all), k)
    preds = {t: v.clone() for t, v in preds.items()}
    grads = {n: g.clone() for n, g in grads.items()}
    losses = {t: v.item() for t, v in losses.items()}
    o_masks, _, _ = O.generate_masks(p, noises, noise_all, k)  # the masks of the full batch: bit-exact against the oracle
    for d, om in zip(cfg["in_domains"], o_masks):
        assert torch.equal(masks[d].cpu(), om), d
    preds2, _, losses2, grads2 = _hip_step(model, cfg, x, (p, noises, noise_all), k)
    assert all(torch.equal(preds[t], preds2[t]) for t in preds) and all(torch.equal(grads[n], grads2[n]) for n in grads)

    gsum = {n: torch.zeros_like(g, dtype=torch.float64) for n, g in grads.items()}
    lsum = {t: 0.0 for t in losses}
    worst_pred = 0.0
    for c in range(B // C):
        sl = slice(c * C, (c + 1) * C)
        xc = {d: t[sl] for d, t in x.items()}
        pc, mc, lc, gc = _hip_step(model, cfg, xc, (p[sl], [n[sl] for n in noises], noise_all[sl]), k)
        for d in mc:
            assert torch.equal(mc[d], masks[d][sl]), (c, d)
        for t in pc:
            worst_pred = max(worst_pred, rel_err(pc[t].float(), preds[t][sl].float()))
            lsum[t] += lc[t].item()
        for n in gc:
            gsum[n] += gc[n].double()
    errs = {f"loss_{t}": abs(lsum[t] / (B // C) - losses[t]) / abs(losses[t]) for t in losses}
    errs["pred"] = worst_pred
    num = sum(((gsum[n] / (B // C) - grads[n].double()) ** 2).sum() for n in grads).sqrt().item()
    den = sum((grads[n].double() ** 2).sum() for n in grads).sqrt().item()
    errs["grad_all"] = num / den
    gerr = {n: rel_err((gsum[n] / (B // C)).float(), grads[n]) for n in grads}
    worst = sorted(gerr.items(), key=lambda kv: -kv[1])[:5]
    print({k_: f"{v:.2e}" for k_, v in errs.items()}, "worst per-tensor grads:", [(n, f"{e:.2e}") for n, e in worst])
    assert all(v < 1e-5 for v in errs.values()), errs
    assert all(e < 1e-4 for e in gerr.values()), worst


def test_step_with_its_own_mask_draws_never_waits_for_the_device():
    """A whole optimisation step with the model drawing its own masks (Dirichlet + uniform noise on the device), under
    torch's synchronisation check: any call that makes the host wait for the device (.item(), a pageable host->device
    copy, a distribution's argument check) raises.  One such wait per step drains the launch queue and leaves the device
    idle at the start of the next step."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from multimae_amd.train import PretrainStep
    cfg = O.small_cfg()
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    model = _build(cfg)
    model.load_state_dict(G.perturbed_state(cfg, seed=7))
    model.cuda()
    step = PretrainStep(model, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, clip_grad=1.0)
    step(x)  # first call: allocations, cached tables
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for _ in range(2):
            loss, _, _ = step(x)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert torch.isfinite(loss).item()


def test_chosen_streams_run_beside_the_current_one():
    """functional.concurrent_stream hands out a stream on another hardware queue than the current stream's (and than the
    ones handed out before): two idle 2-ms kernels queued together must take about the time of one."""
    from multimae_amd import _lib, functional as Fn
    lib = _lib.load()
    assert lib.mmae_spin_us(0, None) != 0  # argument check: the duration must be positive
    dev = torch.device("cuda", torch.cuda.current_device())
    main = torch.cuda.current_stream(dev)
    st = Fn.concurrent_stream(dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    st.wait_stream(main)
    _lib.check(lib.mmae_spin_us(2000, main.cuda_stream), "spin_us")
    _lib.check(lib.mmae_spin_us(2000, st.cuda_stream), "spin_us")
    main.wait_stream(st)
    e1.record(main)
    torch.cuda.synchronize()
    assert e0.elapsed_time(e1) < 3.0, e0.elapsed_time(e1)


@pytest.mark.parametrize("B,k", [(1, 8), (3, 5), (5, 13), (5, 31)])
def test_ragged_batches_and_token_counts_vs_oracle(B, k):
    """Batches and visible-token counts that fill no tile, wavefront or row group evenly (one image; 3 x 6, 5 x 14 and
    5 x 32 encoder rows): outputs, losses and every parameter gradient against the fp32 oracle, masks bit-exact.  With 31 of
    the 32 tokens visible every sample has a task without a masked token: its per-sample loss is 0 / 0, nanmean drops it from
    the loss, and its gradient is NaN in the reference's arithmetic; the step reproduces both."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    cfg = O.small_cfg()
    sd = G.perturbed_state(cfg, seed=11)
    x = G.seeded_inputs(cfg, B=B, seed=5)
    p, noises, noise_all = O.draw_mask_randomness(B, [16, 16], 1.0, torch.Generator().manual_seed(B))
    o_losses, o_preds, o_masks, o_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k)
    model = _build(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    preds, masks, losses, grads = _hip_step(model, cfg, x, (p, noises, noise_all), k)
    for d in masks:
        assert torch.equal(masks[d].cpu(), o_masks[d]), d
    errs = {f"pred_{t}": rel_err(preds[t].float().cpu(), o_preds[t]) for t in preds}
    errs.update({f"loss_{t}": abs(losses[t].item() - float(o_losses[t])) / abs(float(o_losses[t])) for t in preds})
    assert set(grads) == set(o_grads)
    if any(bool(g.isnan().any()) for g in o_grads.values()):
        assert k == 31 and all(torch.isfinite(torch.as_tensor(float(o_losses[t]))) for t in o_losses)
        bad = {n: e for n, e in errs.items() if not e < BF16_TOL}
        assert not bad, bad
        for n in grads:
            assert torch.equal(grads[n].isnan().cpu(), o_grads[n].isnan()), n
        return
    num = sum(((grads[n].cpu().double() - o_grads[n].double()) ** 2).sum() for n in grads).sqrt().item()
    on = torch.sqrt(sum((g.double() ** 2).sum() for g in o_grads.values())).item()
    errs["grad_all"] = num / on
    gerr = {n: rel_err(g.cpu(), o_grads[n]) for n, g in grads.items()}
    worst = sorted(gerr.items(), key=lambda kv: -kv[1])[:4]
    print(B, k, {n: f"{e:.2e}" for n, e in errs.items()}, [(n, f"{e:.2e}") for n, e in worst])
    bad = {n: e for n, e in errs.items() if not e < BF16_TOL}
    assert not bad, bad
    # per tensor: with a few images the sums have few terms to average the bf16 rounding over, and the L1 loss's gradient is a
    # sign (a prediction rounded across its target flips a whole term): twice the bar below four images
    tol = BF16_TOL if B >= 4 else 2 * BF16_TOL
    badg = {n: e for n, e in gerr.items() if not e < tol}
    assert not badg, badg


@pytest.mark.parametrize("num_global", [0, 3])
def test_other_global_token_counts_vs_oracle(num_global):
    """num_global_tokens other than the pretraining configurations' 1 (the reference appends them after the kept tokens and
    hands them to every decoder as context without a position or task embedding): none, and three."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    cfg = dict(O.small_cfg(), num_global=num_global)
    sd = G.perturbed_state(cfg, seed=13)
    assert sd["global_tokens"].shape[1] == num_global
    B, k = 8, 8
    x = G.seeded_inputs(cfg, B=B, seed=6)
    p, noises, noise_all = O.draw_mask_randomness(B, [16, 16], 1.0, torch.Generator().manual_seed(3))
    o_losses, o_preds, o_masks, o_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k)
    model = _build(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    preds, masks, losses, grads = _hip_step(model, cfg, x, (p, noises, noise_all), k)
    for d in masks:
        assert torch.equal(masks[d].cpu(), o_masks[d]), d
    errs = {f"pred_{t}": rel_err(preds[t].float().cpu(), o_preds[t]) for t in preds}
    errs.update({f"loss_{t}": abs(losses[t].item() - float(o_losses[t])) / abs(float(o_losses[t])) for t in preds})
    for n, g in o_grads.items():
        if g.numel():
            assert grads.get(n) is not None, n
            errs["grad_" + n] = rel_err(grads[n].cpu(), g)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(num_global, [(n, f"{e:.2e}") for n, e in worst])
    bad = {n: e for n, e in errs.items() if not e < BF16_TOL}
    assert not bad, bad


def test_other_image_size_than_the_adapters_were_built_for_vs_oracle():
    """Adapters built for 64 x 64 run on 96 x 64 images (6 x 4 tokens per domain): the input position tables are
    resampled bicubically, the decoders' bilinearly (reference: multimae/input_adapters.py:129-131,
    multimae/output_adapters.py:213-218), on a grid that is not square."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    cfg = O.small_cfg()
    sd = G.perturbed_state(cfg, seed=17)
    B, k, H, W = 8, 11, 96, 64  # eight images: the L1 loss's sign gradient needs terms to average bf16 flips over
    gen = torch.Generator().manual_seed(9)
    x = {d: torch.randn(B, C, H, W, generator=gen) for d, C in cfg["in_domains"].items()}
    p, noises, noise_all = O.draw_mask_randomness(B, [24, 24], 1.0, torch.Generator().manual_seed(4))
    o_losses, o_preds, o_masks, o_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k)
    model = _build(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    preds, masks, losses, grads = _hip_step(model, cfg, x, (p, noises, noise_all), k)
    for d in masks:
        assert masks[d].shape == (B, 24) and torch.equal(masks[d].cpu(), o_masks[d]), d
    errs = {f"pred_{t}": rel_err(preds[t].float().cpu(), o_preds[t]) for t in preds}
    errs.update({f"loss_{t}": abs(losses[t].item() - float(o_losses[t])) / abs(float(o_losses[t])) for t in preds})
    errs.update({"grad_" + n: rel_err(grads[n].cpu(), g) for n, g in o_grads.items()})
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print([(n, f"{e:.2e}") for n, e in worst])
    assert all(preds[t].shape[-2:] == (H, W) for t in preds)
    bad = {n: e for n, e in errs.items() if not e < BF16_TOL}
    assert not bad, bad


def test_mask_inputs_false_encodes_every_token_vs_oracle():
    """forward(mask_inputs=False): every token is kept (in the shuffled order of the mask draw), the masks are all zero
    (reference: multimae/multimae.py:366-372)."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    cfg = O.small_cfg()
    sd = G.perturbed_state(cfg, seed=19)
    B = 4
    x = G.seeded_inputs(cfg, B=B, seed=8)
    rnd = O.draw_mask_randomness(B, [16, 16], 1.0, torch.Generator().manual_seed(5))
    o_preds, o_masks = O.forward(sd, cfg, x, *rnd, 32)
    model = _build(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda().eval()
    with torch.no_grad():
        preds, masks = model({d: t.cuda() for d, t in x.items()}, mask_inputs=False, num_encoded_tokens=8, mask_randomness=rnd)
    for d in masks:
        assert not masks[d].any() and torch.equal(masks[d].cpu(), o_masks[d]), d
    errs = {t: rel_err(preds[t].float().cpu(), o_preds[t]) for t in preds}
    print({t: f"{e:.2e}" for t, e in errs.items()})
    assert all(e < BF16_TOL for e in errs.values()), errs


def test_pretrain_step_loss_on_unmasked_vs_oracle():
    """PretrainStep(loss_on_unmasked=True): the task losses are taken over every token (the reference's train_one_epoch
    passes no mask then, run_pretraining_multimae.py:491-496); losses and gradients against the oracle's arithmetic."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from multimae_amd.train import PretrainStep
    cfg = O.small_cfg()
    sd = G.perturbed_state(cfg, seed=23)
    B, k = 8, 8
    x = G.seeded_inputs(cfg, B=B, seed=4)
    rnd = O.draw_mask_randomness(B, [16, 16], 1.0, torch.Generator().manual_seed(6))
    work = {n: v.detach().clone().requires_grad_(not n.endswith(O.FROZEN_SUFFIX)) for n, v in sd.items()}
    o_preds, _ = O.forward(work, cfg, x, *rnd, k)
    o_losses = {name: fn(o_preds[name].float(), x[tgt], mask=None, **kw) for name, (fn, tgt, mk, kw) in O.default_loss_table(cfg).items()}
    names = [n for n in work if work[n].requires_grad]
    o_grads = dict(zip(names, torch.autograd.grad(sum(o_losses.values()), [work[n] for n in names])))

    model = _build(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    step = PretrainStep(model, _losses(cfg), num_encoded_tokens=k, lr=0.0, weight_decay=0.0, loss_on_unmasked=True, keep_grads=True)
    loss, task_losses, _ = step({d: t.cuda() for d, t in x.items()}, mask_randomness=rnd)
    errs = {f"loss_{t}": abs(task_losses[t].item() - float(o_losses[t].detach())) / abs(float(o_losses[t].detach())) for t in o_losses}
    errs.update({"grad_" + n: rel_err(p.grad.cpu(), o_grads[n]) for n, p in model.named_parameters() if p.requires_grad})
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print([(n, f"{e:.2e}") for n, e in worst])
    bad = {n: e for n, e in errs.items() if not e < BF16_TOL}
    assert not bad, bad


def test_image_domain_at_another_stride_level_vs_oracle():
    """A plain image domain at stride_level 2 (depth at half resolution: 8 x 8 patches on the same 4 x 4 token grid; reference:
    multimae/input_adapters.py:64-65, multimae/output_adapters.py:101-102) as input and as output, forward and backward."""
    import multimae_amd as M
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    out = {"rgb": (3, "rgb"), "depth": (1, "depth")}
    cfg = O.make_cfg({"rgb": 3, "depth": 1}, out, dim=128, depth=2, heads=4, dec_dim=64, dec_depth=1, dec_heads=2, patch=16,
                     image=64, stride={"depth": 2})
    sd = G.perturbed_state(cfg, seed=29)
    B, k = 8, 8
    gen = torch.Generator().manual_seed(12)
    x = {"rgb": torch.randn(B, 3, 64, 64, generator=gen), "depth": torch.randn(B, 1, 32, 32, generator=gen)}
    rnd = O.draw_mask_randomness(B, [16, 16], 1.0, torch.Generator().manual_seed(7))
    work = {n: v.detach().clone().requires_grad_(not n.endswith(O.FROZEN_SUFFIX)) for n, v in sd.items()}
    o_preds, o_masks = O.forward(work, cfg, x, *rnd, k)
    assert o_preds["depth"].shape == (B, 1, 32, 32)
    names = [n for n in work if work[n].requires_grad]
    o_grads = dict(zip(names, torch.autograd.grad(sum(p.square().mean() for p in o_preds.values()), [work[n] for n in names])))

    ia = {d: M.PatchedInputAdapter(num_channels=C, stride_level=cfg["stride"].get(d, 1), patch_size_full=16, image_size=64)
          for d, C in cfg["in_domains"].items()}
    oa = {d: M.SpatialOutputAdapter(num_channels=C, stride_level=cfg["stride"].get(d, 1), patch_size_full=16, dim_tokens=64, depth=1,
                                    num_heads=2, task=t, context_tasks=list(cfg["in_domains"]), image_size=64)
          for d, (C, t) in out.items()}
    model = M.MultiMAE(ia, oa, num_global_tokens=1, dim_tokens=128, depth=2, num_heads=4)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    preds, masks = model({d: t.cuda() for d, t in x.items()}, num_encoded_tokens=k, alphas=1.0, mask_randomness=rnd)
    sum(p.float().square().mean() for p in preds.values()).backward()
    M.functional.join_side_stream()
    for d in masks:
        assert torch.equal(masks[d].cpu(), o_masks[d]), d
    errs = {f"pred_{t}": rel_err(preds[t].float().cpu(), o_preds[t].detach()) for t in preds}
    errs.update({"grad_" + n: rel_err(p.grad.cpu(), o_grads[n]) for n, p in model.named_parameters() if p.requires_grad})
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print([(n, f"{e:.2e}") for n, e in worst])
    bad = {n: e for n, e in errs.items() if not e < BF16_TOL}
    assert not bad, bad


@pytest.mark.parametrize("domains", [{"rgb": 3}, {"rgb": 3, "depth": 1, "normal": 3, "edges": 1, "extra": 2}])
def test_one_and_five_domains_vs_oracle(domains):
    """Domain counts other than the pretraining configurations' two or three: a single-domain masked autoencoder, and five
    image domains (the Dirichlet draw, the joint shuffle and every decoder's context span all of them)."""
    import multimae_amd as M
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    out = {d: (C, d) for d, C in domains.items()}
    cfg = O.make_cfg(domains, out, dim=128, depth=2, heads=4, dec_dim=64, dec_depth=1, dec_heads=2, patch=16, image=64)
    sd = G.perturbed_state(cfg, seed=31)
    B, T = 8, len(domains)
    k = 4 * T + 1
    gen = torch.Generator().manual_seed(14)
    x = {d: torch.randn(B, C, 64, 64, generator=gen) for d, C in domains.items()}
    rnd = O.draw_mask_randomness(B, [16] * T, 1.0, torch.Generator().manual_seed(8))
    work = {n: v.detach().clone().requires_grad_(not n.endswith(O.FROZEN_SUFFIX)) for n, v in sd.items()}
    o_preds, o_masks = O.forward(work, cfg, x, *rnd, k)
    o_losses = {d: O.masked_mse(o_preds[d], x[d], mask=o_masks[d], patch_size=16) for d in domains}
    names = [n for n in work if work[n].requires_grad]
    o_grads = dict(zip(names, torch.autograd.grad(sum(o_losses.values()), [work[n] for n in names])))

    ia = {d: M.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=16, image_size=64) for d, C in domains.items()}
    oa = {d: M.SpatialOutputAdapter(num_channels=C, stride_level=1, patch_size_full=16, dim_tokens=64, depth=1, num_heads=2, task=d,
                                    context_tasks=list(domains), image_size=64) for d, C in domains.items()}
    model = M.MultiMAE(ia, oa, num_global_tokens=1, dim_tokens=128, depth=2, num_heads=4)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    xg = {d: t.cuda() for d, t in x.items()}
    preds, masks = model(xg, num_encoded_tokens=k, alphas=1.0, mask_randomness=rnd)
    losses = {d: M.MaskedMSELoss(patch_size=16, stride=1)(preds[d], xg[d], mask=masks[d]) for d in domains}
    sum(losses.values()).backward()
    M.functional.join_side_stream()
    for d in masks:
        assert torch.equal(masks[d].cpu(), o_masks[d]), d
    errs = {f"pred_{d}": rel_err(preds[d].float().cpu(), o_preds[d].detach()) for d in preds}
    errs.update({f"loss_{d}": abs(losses[d].item() - float(o_losses[d].detach())) / abs(float(o_losses[d].detach())) for d in domains})
    errs.update({"grad_" + n: rel_err(p.grad.cpu(), o_grads[n]) for n, p in model.named_parameters() if p.requires_grad})
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(T, [(n, f"{e:.2e}") for n, e in worst])
    bad = {n: e for n, e in errs.items() if not e < BF16_TOL}
    assert not bad, bad


def test_no_qkv_bias_and_another_mlp_ratio_vs_oracle():
    """qkv_bias=False (no bias on qkv / q / kv anywhere; the kernels then add a zero vector) and mlp_ratio=2 in the encoder
    and the decoders, forward and backward."""
    import multimae_amd as M
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    out = {"rgb": (3, "rgb"), "depth": (1, "depth")}
    cfg = O.make_cfg({"rgb": 3, "depth": 1}, out, dim=128, depth=2, heads=4, dec_dim=64, dec_depth=1, dec_heads=2, patch=16,
                     image=64, mlp_ratio=2.0)
    sd = {n: v for n, v in G.perturbed_state(cfg, seed=37).items() if not n.endswith((".qkv.bias", ".q.bias", ".kv.bias"))}
    assert sd["encoder.0.mlp.fc1.weight"].shape == (256, 128)
    B, k = 8, 8
    x = G.seeded_inputs(cfg, B=B, seed=10)
    rnd = O.draw_mask_randomness(B, [16, 16], 1.0, torch.Generator().manual_seed(9))
    work = {n: v.detach().clone().requires_grad_(not n.endswith(O.FROZEN_SUFFIX)) for n, v in sd.items()}
    o_preds, o_masks = O.forward(work, cfg, x, *rnd, k)
    o_losses = {d: O.masked_mse(o_preds[d], x[d], mask=o_masks[d], patch_size=16) for d in out}
    names = [n for n in work if work[n].requires_grad]
    o_grads = dict(zip(names, torch.autograd.grad(sum(o_losses.values()), [work[n] for n in names])))

    ia = {d: M.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=16, image_size=64) for d, C in cfg["in_domains"].items()}
    oa = {d: M.SpatialOutputAdapter(num_channels=C, stride_level=1, patch_size_full=16, dim_tokens=64, depth=1, num_heads=2, task=t,
                                    context_tasks=list(cfg["in_domains"]), image_size=64, qkv_bias=False, mlp_ratio=2.0)
          for d, (C, t) in out.items()}
    model = M.MultiMAE(ia, oa, num_global_tokens=1, dim_tokens=128, depth=2, num_heads=4, qkv_bias=False, mlp_ratio=2.0)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    xg = {d: t.cuda() for d, t in x.items()}
    preds, masks = model(xg, num_encoded_tokens=k, alphas=1.0, mask_randomness=rnd)
    losses = {d: M.MaskedMSELoss(patch_size=16, stride=1)(preds[d], xg[d], mask=masks[d]) for d in out}
    sum(losses.values()).backward()
    M.functional.join_side_stream()
    errs = {f"pred_{d}": rel_err(preds[d].float().cpu(), o_preds[d].detach()) for d in preds}
    errs.update({f"loss_{d}": abs(losses[d].item() - float(o_losses[d].detach())) / abs(float(o_losses[d].detach())) for d in out})
    errs.update({"grad_" + n: rel_err(p.grad.cpu(), o_grads[n]) for n, p in model.named_parameters() if p.requires_grad})
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print([(n, f"{e:.2e}") for n, e in worst])
    bad = {n: e for n, e in errs.items() if not e < BF16_TOL}
    assert not bad, bad


@pytest.mark.parametrize("dec_depth", [0, 3])
def test_decoders_without_and_with_three_transformer_blocks_vs_oracle(dec_depth):
    """SpatialOutputAdapter(depth=0) (the class's default: cross-attention and MLP only) and depth=3."""
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    cfg = dict(O.small_cfg(), dec_depth=dec_depth)
    sd = G.perturbed_state(cfg, seed=41)
    B, k = 8, 8
    x = G.seeded_inputs(cfg, B=B, seed=12)
    p, noises, noise_all = O.draw_mask_randomness(B, [16, 16], 1.0, torch.Generator().manual_seed(10))
    o_losses, o_preds, o_masks, o_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k)
    model = _build(cfg)
    assert (len(model.output_adapters["rgb"].decoder_transformer) if dec_depth else
            isinstance(model.output_adapters["rgb"].decoder_transformer, torch.nn.Identity))
    model.load_state_dict(sd, strict=True)
    model.cuda()
    preds, masks, losses, grads = _hip_step(model, cfg, x, (p, noises, noise_all), k)
    errs = {f"pred_{t}": rel_err(preds[t].float().cpu(), o_preds[t]) for t in preds}
    errs.update({f"loss_{t}": abs(losses[t].item() - float(o_losses[t])) / abs(float(o_losses[t])) for t in preds})
    assert set(grads) == set(o_grads)
    errs.update({"grad_" + n: rel_err(g.cpu(), o_grads[n]) for n, g in grads.items()})
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(dec_depth, [(n, f"{e:.2e}") for n, e in worst])
    bad = {n: e for n, e in errs.items() if not e < BF16_TOL}
    assert not bad, bad
