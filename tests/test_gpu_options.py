"""The module settings outside the pretraining configuration against the reference's golden outputs
(tests/golden/options.npz, written by make_golden.py gen_options from the reference's own classes): decoders without
cross-attention, decoders with their own queries, a decoder whose task is not an input, learnable input position tables,
stochastic depth in training mode.  bf16 tolerance: 2e-2 relative (L2 over a tensor); masks bit-exact."""
import numpy as np
import pytest
import torch

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
BF16_TOL = 2e-2


def _build(case):
    import multimae_amd as M
    from tests.golden import common as G
    cfg, ia_kw, oa_kw, m_kw, training = G.option_cases()[case]
    S = cfg["image"]
    ia = {d: M.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"], image_size=S, **ia_kw.get(d, {}))
          for d, C in cfg["in_domains"].items()}
    oa = {name: M.SpatialOutputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"], dim_tokens=cfg["dec_dim"],
                                       depth=cfg["dec_depth"], num_heads=cfg["dec_heads"], task=task,
                                       context_tasks=list(cfg["in_domains"]), image_size=S, **oa_kw)
          for name, (C, task) in cfg["out_domains"].items()}
    model = M.MultiMAE(ia, oa, num_global_tokens=cfg["num_global"], dim_tokens=cfg["dim"], depth=cfg["depth"],
                       num_heads=cfg["heads"], **m_kw)
    tables = [f"input_adapters.{d}.pos_emb" for d, kw in ia_kw.items() if kw.get("sincos_pos_emb") is False]
    # strict: the variant has exactly the reference's state_dict keys (no cross-attention modules without use_xattn)
    model.load_state_dict(G.option_state(cfg, 7, list(model.state_dict().keys()), tables), strict=True)
    return model.cuda().train(training), cfg


@pytest.mark.parametrize("case", ["noxattn", "ownq", "rgbonly", "learnpos", "droppath"])
def test_module_option_vs_reference_golden(case, monkeypatch):
    import multimae_amd as M
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    z = G.load("options.npz")
    pre = case + "_"
    model, cfg = _build(case)
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == int(z[pre + "n_trainable"])
    x_all = {d: t.cuda() for d, t in G.seeded_inputs(O.small_cfg(), B=G.OPTION_BATCH, seed=3).items()}
    x = {d: x_all[d] for d in cfg["in_domains"]}
    rnd = (torch.from_numpy(z[pre + "p"]), [torch.from_numpy(z[pre + f"noise{i}"]) for i in range(len(x))],
           torch.from_numpy(z[pre + "noise_all"]))
    if pre + "droppath_draws" in z.files:  # replay the reference's stochastic-depth draws, in call order
        draws, real_rand = list(torch.from_numpy(z[pre + "droppath_draws"])), torch.rand

        def replayed_rand(*size, **kw):
            shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
            if shape == (G.OPTION_BATCH, 1, 1):
                return draws.pop(0).to(device=kw.get("device"), dtype=kw.get("dtype") or torch.float32)
            return real_rand(*size, **kw)
        monkeypatch.setattr(torch, "rand", replayed_rand)
    losses_fn = {"rgb": M.MaskedMSELoss(patch_size=16, stride=1), "depth": M.MaskedMSELoss(patch_size=16, stride=1)}
    preds, masks = model(x, num_encoded_tokens=8, alphas=1.0, mask_randomness=rnd)
    losses = {t: losses_fn[t](preds[t], x_all[t], mask=masks.get(t)) for t in preds}
    sum(losses.values()).backward()
    if pre + "droppath_draws" in z.files:
        assert not draws, f"{len(draws)} stochastic-depth draws of the reference were not consumed"
    for d in masks:
        assert np.array_equal(masks[d].cpu().numpy(), z[pre + f"mask_{d}"].astype(np.int64)), d
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    report = {}
    for t in preds:
        report[f"pred_{t}"] = rel_err(preds[t].float().cpu(), z[pre + f"pred_{t}"].astype(np.float32))
        report[f"loss_{t}"] = abs(losses[t].item() - float(z[pre + f"loss_{t}"])) / abs(float(z[pre + f"loss_{t}"]))
    checked = 0
    for key in z.files:
        if key.startswith(pre + "grad_") and key != pre + "grad_norm":
            name = key[len(pre) + 5:]
            assert grads.get(name) is not None, f"no gradient for {name}"
            report["grad_" + name] = rel_err(grads[name].cpu(), z[key])
            checked += 1
    assert checked >= 10
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values() if g is not None)).item()
    report["grad_norm"] = abs(gn - float(z[pre + "grad_norm"])) / float(z[pre + "grad_norm"])
    print(case, {k_: f"{v:.2e}" for k_, v in report.items()})
    bad = {k_: v for k_, v in report.items() if not v < BF16_TOL}
    assert not bad, bad


def test_dropout_settings_are_the_identity_in_eval_mode_and_drop_in_training_mode():
    """drop_rate acts on the projections' outputs (the reference's proj_drop / Mlp.drop): eval mode equals the model
    without dropout bitwise; training mode zeroes about drop_rate of a block's branch outputs; dropout on the attention
    probabilities runs through the materialised form (next test)."""
    import multimae_amd as M
    torch.manual_seed(0)
    blk0 = M.multimae_utils.Block(128, 4, qkv_bias=True, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6)).cuda()
    blk = M.multimae_utils.Block(128, 4, qkv_bias=True, drop=0.25, attn_drop=0.1, drop_path=0.2,
                                 norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6)).cuda()
    blk.load_state_dict(blk0.state_dict(), strict=True)
    x = torch.randn(4, 40, 128, device="cuda")
    assert torch.equal(blk.eval()(x), blk0.eval()(x))
    y = blk.train()(x.requires_grad_())
    y.sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(x.grad).all() and blk.mlp.fc1.weight.grad is not None
    mlp = M.multimae_utils.Mlp(128, 512, drop=0.25).cuda().train()
    frac = (mlp(torch.randn(64, 40, 128, device="cuda")) == 0).float().mean().item()
    assert 0.2 < frac < 0.3, frac


@pytest.mark.parametrize("cross", [False, True])
def test_dropout_on_the_attention_probabilities_in_training_mode(cross):
    """attn_drop > 0 in training mode (reference: multimae/multimae_utils.py:200-204, :240-244): one dropout call on the
    [B,H,Nq,Nk] probabilities between softmax and the value product.  Checked against the same arithmetic written out in fp32
    with the same seed (the same call on a tensor of the same shape draws the same mask): output and input gradients within the
    bf16 tolerance of the projections, about attn_drop of the probabilities dropped, eval mode equal to the module without
    dropout."""
    import torch.nn.functional as F
    import multimae_amd as M
    torch.manual_seed(1)
    B, N, Mctx, D, H, p = 3, 40, 23, 128, 4, 0.2
    mod = (M.multimae_utils.CrossAttention if cross else M.multimae_utils.Attention)(D, num_heads=H, qkv_bias=True, attn_drop=p).cuda()
    x = torch.randn(B, N, D, device="cuda", requires_grad=True)
    ctx = torch.randn(B, Mctx, D, device="cuda", requires_grad=True) if cross else None
    args = (x, ctx) if cross else (x,)

    def written_out(x, ctx):
        d = D // H
        if cross:
            q = F.linear(x, mod.q.weight, mod.q.bias).reshape(B, N, H, d).permute(0, 2, 1, 3)
            k, v = F.linear(ctx, mod.kv.weight, mod.kv.bias).reshape(B, Mctx, 2, H, d).permute(2, 0, 3, 1, 4).unbind(0)
        else:
            q, k, v = F.linear(x, mod.qkv.weight, mod.qkv.bias).reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4).unbind(0)
        attn = F.dropout(((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(-1), p, True)
        dropped = (attn == 0).float().mean().item()
        return F.linear((attn @ v).transpose(1, 2).reshape(B, N, D), mod.proj.weight, mod.proj.bias), dropped

    g = torch.randn(B, N, D, device="cuda")
    torch.manual_seed(7)
    y = mod.train()(*args)
    y.backward(g)
    got = [y.detach().clone()] + [t.grad.clone() for t in args] + [mod.proj.weight.grad.clone()]
    for t in args:
        t.grad = None
    mod.proj.weight.grad = None
    torch.manual_seed(7)
    yr, dropped = written_out(*((x, ctx) if cross else (x, None)))
    yr.backward(g)
    want = [yr.detach()] + [t.grad for t in args] + [mod.proj.weight.grad]
    assert 0.15 < dropped < 0.25, dropped
    for a, b in zip(got, want):
        assert rel_err(a, b) < BF16_TOL, rel_err(a, b)
    plain = (M.multimae_utils.CrossAttention if cross else M.multimae_utils.Attention)(D, num_heads=H, qkv_bias=True).cuda()
    plain.load_state_dict(mod.state_dict())
    assert torch.equal(mod.eval()(*args), plain.eval()(*args))


def test_learnable_output_position_table_is_refused_as_in_the_reference():
    import multimae_amd as M
    with pytest.raises(NotImplementedError):
        M.SpatialOutputAdapter(num_channels=3, stride_level=1, patch_size_full=16, learnable_pos_emb=True)


def test_learnable_position_tables_in_flat_storage_take_the_same_gradient():
    """With the parameters in flat storage (the step's layout) the position-table gradient is written in place by the
    embedding's backward; it must equal the one autograd delivers for the plain module."""
    import multimae_amd as M
    from multimae_amd.train import FlatParams
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    z = G.load("options.npz")
    x = {d: t.cuda() for d, t in G.seeded_inputs(O.small_cfg(), B=G.OPTION_BATCH, seed=3).items()}
    rnd = (torch.from_numpy(z["learnpos_p"]), [torch.from_numpy(z[f"learnpos_noise{i}"]) for i in range(2)],
           torch.from_numpy(z["learnpos_noise_all"]))
    grads = []
    for flat in (False, True):
        model, _ = _build("learnpos")
        holder = FlatParams(model) if flat else None
        preds, masks = model(x, num_encoded_tokens=8, alphas=1.0, mask_randomness=rnd)
        sum(M.MaskedMSELoss(patch_size=16, stride=1)(preds[t], x[t], mask=masks[t]) for t in preds).backward()
        M.functional.join_side_stream()
        grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad})
        if flat:
            assert all(p.grad.data_ptr() >= holder.grads.data_ptr() for p in model.parameters() if p.requires_grad)
    for name in ("input_adapters.rgb.pos_emb", "input_adapters.depth.pos_emb", "global_tokens"):
        assert grads[0][name].abs().sum() > 0
        assert torch.equal(grads[0][name], grads[1][name]), name


@pytest.mark.parametrize("eps", [1e-5, 0.5])
def test_layer_norm_eps_comes_from_the_norm_layer(eps, monkeypatch):
    """norm_layer with another eps than the pretraining configurations' 1e-6 (the reference's Block itself defaults to
    nn.LayerNorm, eps 1e-5): encoder blocks, decoder head and decoder blocks against the oracle run with that eps.
    At eps = 0.5 every normalised row shrinks by more than a tenth, so a kernel that ignored the setting fails the bar."""
    from functools import partial
    import multimae_amd as M
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from tests.test_gpu_model import _hip_step
    cfg = O.small_cfg()
    sd = G.perturbed_state(cfg, seed=7)
    x = G.seeded_inputs(cfg, B=4, seed=3)
    k = 8
    p, noises, noise_all = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    base_losses, base_preds, _, _ = O.step_grads(sd, cfg, x, p, noises, noise_all, k)
    monkeypatch.setattr(O, "LN_EPS", eps)
    o_losses, o_preds, o_masks, o_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k)
    if eps == 0.5:
        assert rel_err(o_preds["rgb"], base_preds["rgb"]) > 0.05

    norm = partial(torch.nn.LayerNorm, eps=eps)
    S = cfg["image"]
    ia = {d: M.PatchedInputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"], image_size=S)
          for d, C in cfg["in_domains"].items()}
    oa = {name: M.SpatialOutputAdapter(num_channels=C, stride_level=1, patch_size_full=cfg["patch"], dim_tokens=cfg["dec_dim"],
                                       depth=cfg["dec_depth"], num_heads=cfg["dec_heads"], task=task, norm_layer=norm,
                                       context_tasks=list(cfg["in_domains"]), image_size=S)
          for name, (C, task) in cfg["out_domains"].items()}
    model = M.MultiMAE(ia, oa, num_global_tokens=cfg["num_global"], dim_tokens=cfg["dim"], depth=cfg["depth"],
                       num_heads=cfg["heads"], norm_layer=norm)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    preds, masks, losses, grads = _hip_step(model, cfg, x, (p, noises, noise_all), k)
    errs = {f"pred_{t}": rel_err(preds[t].float().cpu(), o_preds[t]) for t in preds}
    errs.update({f"loss_{t}": abs(losses[t].item() - float(o_losses[t])) / abs(float(o_losses[t])) for t in preds})
    errs.update({"grad_" + n: rel_err(g.cpu(), o_grads[n]) for n, g in grads.items()})
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print(eps, [(n, f"{e:.2e}") for n, e in worst])
    bad = {n: e for n, e in errs.items() if not e < BF16_TOL}
    assert not bad, bad


def test_block_with_a_norm_module_that_is_not_an_affine_layer_norm_runs_it_as_a_module(monkeypatch):
    """Any norm_layer is accepted: what the LayerNorm kernel does not serve (here: LayerNorm without weight and bias)
    runs as the torch module between the attention and MLP nodes."""
    from functools import partial
    import multimae_amd as M
    from oracle import multimae_oracle as O
    torch.manual_seed(1)
    D, heads = 128, 4
    blk = M.multimae_utils.Block(D, heads, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, elementwise_affine=False)).cuda()
    assert not blk._ln_kernel and "norm1.weight" not in blk.state_dict()
    x = torch.randn(4, 40, D, device="cuda", requires_grad=True)
    y = blk(x)
    y.square().mean().backward()
    sd = {"b." + n: t.detach().cpu() for n, t in blk.state_dict().items()}
    for n in ("norm1", "norm2"):
        sd[f"b.{n}.weight"], sd[f"b.{n}.bias"] = torch.ones(D), torch.zeros(D)
    monkeypatch.setattr(O, "LN_EPS", 1e-5)
    xr = x.detach().cpu().requires_grad_()
    yr = O.block(xr, sd, "b", heads)
    yr.square().mean().backward()
    assert rel_err(y.detach().cpu(), yr.detach()) < BF16_TOL
    assert rel_err(x.grad.cpu(), xr.grad) < BF16_TOL


def test_block_with_another_activation_runs_it_between_the_two_products():
    """act_layer other than nn.GELU (whose exact form the fc1 epilogue computes): the activation module runs between two
    GEMM nodes; checked against the same block in fp32 torch math on the host.  (A smooth activation: with ReLU the bf16
    rounding of a pre-activation next to zero flips its derivative, 2-3 % of the fc1 gradient at this size.)"""
    from functools import partial
    import multimae_amd as M
    from oracle import multimae_oracle as O
    torch.manual_seed(2)
    D, heads = 128, 4
    blk = M.multimae_utils.Block(D, heads, qkv_bias=True, act_layer=torch.nn.SiLU,
                                 norm_layer=partial(torch.nn.LayerNorm, eps=1e-6)).cuda()
    assert set(blk.state_dict()) == {f"{m}.{p}" for m in ("norm1", "norm2", "attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")
                                     for p in ("weight", "bias")}
    x = torch.randn(4, 40, D, device="cuda", requires_grad=True)
    y = blk(x)
    y.square().mean().backward()
    M.functional.join_side_stream()  # the weight gradients are formed on the second stream
    sd = {"b." + n: t.detach().cpu().requires_grad_() for n, t in blk.state_dict().items()}
    xr = x.detach().cpu().requires_grad_()
    h = xr + O.self_attention(O._ln(xr, sd, "b.norm1"), sd, "b.attn", heads)
    yr = h + O._lin(torch.nn.functional.silu(O._lin(O._ln(h, sd, "b.norm2"), sd, "b.mlp.fc1")), sd, "b.mlp.fc2")
    yr.square().mean().backward()
    errs = {"y": rel_err(y.detach().cpu(), yr.detach()), "dx": rel_err(x.grad.cpu(), xr.grad)}
    errs.update({n: rel_err(p.grad.cpu(), sd["b." + n].grad) for n, p in blk.named_parameters()})
    print({n: f"{e:.2e}" for n, e in errs.items()})
    bad = {n: e for n, e in errs.items() if not e < BF16_TOL}
    assert not bad, bad
