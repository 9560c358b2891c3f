"""The oracle (oracle/multimae_oracle.py) against the fixtures the reference produced
(tests/golden/*.npz, written by tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import multimae_oracle as O
from tests.conftest import rel_err
from tests.golden import common as G

FP32_TOL = 1e-4  # relative, the fp32 bar of the parity contract


def test_masks_bit_exact():
    z = G.load("masks.npz")
    n = int(z["n_cases"])
    assert n >= 18
    rounding_cases = 0
    for c in range(n):
        pre = f"c{c}_"
        k = int(z[pre + "k"])
        p = torch.from_numpy(z[pre + "p"])
        T = p.shape[1]
        noises = [torch.from_numpy(z[pre + f"noise{i}"]) for i in range(T)]
        noise_all = torch.from_numpy(z[pre + "noise_all"])
        masks, ids_keep, ids_restore = O.generate_masks(p, noises, noise_all, k)
        assert np.array_equal(ids_keep.numpy(), z[pre + "ids_keep"].astype(np.int64)), c
        assert np.array_equal(ids_restore.numpy(), z[pre + "ids_restore"].astype(np.int64)), c
        assert np.array_equal(torch.cat(masks, 1).numpy(), z[pre + "mask"].astype(np.int64)), c
        assert ids_keep.dtype == torch.int64 and masks[0].dtype == torch.int64
        rounding_cases += int((O.samples_per_task(p, k).sum(1) != k).any())
    assert rounding_cases > 0, "fixtures must include rows where the per-task budgets do not sum to k"


def test_posemb():
    z = G.load("posemb.npz")
    for (h, w, d) in [(14, 14, 768), (14, 14, 256), (3, 5, 8), (24, 24, 1024)]:
        t = O.sincos_posemb(h, w, d)
        tag = f"{h}x{w}x{d}"
        assert tuple(t.shape) == (1, d, h, w)
        flat = t.flatten()
        idx = torch.from_numpy(z[tag + "_probe_idx"])
        np.testing.assert_allclose(flat[idx].numpy(), z[tag + "_probe_val"], rtol=0, atol=2e-6)
        if tag + "_full" in z:
            np.testing.assert_allclose(t.numpy(), z[tag + "_full"], rtol=0, atol=2e-6)
    # row/column quirk: the first channel quarter varies along image rows
    t = O.sincos_posemb(14, 14, 768)
    assert abs(t[0, 0, 1, 0].item() - 0.8414709568) < 1e-6 and t[0, 0, 0, 1].item() == 0.0


def _tiny_inputs():
    z = G.load("tiny_e2e.npz")
    cfg = O.tiny_cfg()
    sd = G.perturbed_state(cfg, seed=7)
    x = G.seeded_inputs(cfg, B=2, seed=3)
    p = torch.from_numpy(z["p"])
    noises = [torch.from_numpy(z["noise0"]), torch.from_numpy(z["noise1"])]
    noise_all = torch.from_numpy(z["noise_all"])
    return z, cfg, sd, x, p, noises, noise_all, int(z["k"])


def test_tiny_end_to_end_fp32():
    z, cfg, sd, x, p, noises, noise_all, k = _tiny_inputs()
    assert len(sd) == int(z["n_state_keys"])
    losses, preds, masks, grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k)
    for d in masks:
        assert np.array_equal(masks[d].numpy(), z[f"mask_{d}"].astype(np.int64))
    for t in preds:
        assert rel_err(preds[t], z[f"pred_{t}"]) < FP32_TOL, t
        assert abs(float(losses[t]) - float(z[f"loss_{t}"])) / abs(float(z[f"loss_{t}"])) < FP32_TOL, t
    for name in G.TINY_GRAD_NAMES:
        assert rel_err(grads[name], z["grad_" + name]) < FP32_TOL, name
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values() if g is not None))
    assert abs(gn.item() - float(z["grad_norm"])) / float(z["grad_norm"]) < FP32_TOL
    assert all(g is not None for g in grads.values()), "every trainable tensor receives a gradient"


def test_three_domain_semseg_end_to_end_fp32():
    """rgb + depth + class map at stride 4: three-domain masking, stride-4 decoder output and masked CE are the
    reference's; only the class-map input adapter (absent from the reference fork) is the fixture script's own."""
    z = G.load("small_semseg_e2e.npz")
    cfg = O.small_semseg_cfg()
    sd = G.perturbed_state(cfg, seed=9)
    x = G.seeded_inputs(cfg, B=2, seed=4)
    noises = [torch.from_numpy(z[f"noise{i}"]) for i in range(3)]
    losses, preds, masks, grads = O.step_grads(sd, cfg, x, torch.from_numpy(z["p"]), noises, torch.from_numpy(z["noise_all"]),
                                               int(z["k"]))
    for d in masks:
        assert np.array_equal(masks[d].numpy(), z[f"mask_{d}"].astype(np.int64))
    assert preds["semseg"].shape == (2, 9, 16, 16)
    for t in preds:
        assert rel_err(preds[t], z[f"pred_{t}"]) < FP32_TOL, t
        assert abs(float(losses[t]) - float(z[f"loss_{t}"])) / abs(float(z[f"loss_{t}"])) < FP32_TOL, t
    for name in G.SEMSEG_GRAD_NAMES:
        assert rel_err(grads[name], z["grad_" + name]) < FP32_TOL, name
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values() if g is not None))
    assert abs(gn.item() - float(z["grad_norm"])) / float(z["grad_norm"]) < FP32_TOL


def test_tiny_end_to_end_bf16_autocast():
    z, cfg, sd, x, p, noises, noise_all, k = _tiny_inputs()
    losses, preds, _ = O.step_losses(sd, cfg, x, p, noises, noise_all, k, autocast_dtype=torch.bfloat16)
    for t in losses:
        ref = float(z[f"bf16_loss_{t}"])
        assert abs(float(losses[t]) - ref) / abs(ref) < 2e-2, t
    assert preds["depth"].dtype == torch.bfloat16
    assert rel_err(preds["depth"].float(), z["bf16_pred_depth"]) < 2e-2


LOSS_FN = {"mse": (O.masked_mse, {}), "l1": (O.masked_l1, {}),
           "msenp": (O.masked_mse, {"norm_pix": True}), "l1np": (O.masked_l1, {"norm_pix": True})}


@pytest.mark.parametrize("tagc", ["c3", "c1"])
def test_losses(tagc):
    z = G.load("losses.npz")

    def check(tag, fn, kw, mask_key=True):
        pred = torch.from_numpy(z[tag + "_pred"]).requires_grad_(True)
        target = torch.from_numpy(z[tag + "_target"])
        mask = torch.from_numpy(z[tag + "_mask"].astype(np.int64)) if (tag + "_mask") in z else None
        loss = fn(pred, target, mask=mask, **kw)
        ref = z[tag + "_loss"]
        if ref.dtype == np.int64:  # "nothing masked" early return
            assert loss.dtype == torch.int64 and int(loss) == int(ref)
            return
        assert abs(float(loss.detach()) - float(ref)) <= FP32_TOL * abs(float(ref)), tag
        loss.backward()
        g, gref = pred.grad, torch.from_numpy(z[tag + "_dpred"])
        # a sample with no masked token has a 0/0 in its backward: the reference yields NaN there
        assert torch.equal(g.isnan(), gref.isnan()), tag
        ok = ~gref.isnan()
        assert rel_err(g[ok], gref[ok]) < FP32_TOL, tag

    for nm, (fn, kw) in LOSS_FN.items():
        check(f"{nm}_{tagc}", fn, kw)
    check(f"mse_rowzero_{tagc}", O.masked_mse, {})
    check(f"mse_allzero_{tagc}", O.masked_mse, {})
    check(f"mse_nomask_{tagc}", O.masked_mse, {})
    check(f"l1_nomask_{tagc}", O.masked_l1, {})


def test_cross_entropy():
    z = G.load("losses.npz")
    for tag, kw in (("ce", {}), ("ce_ls", {"label_smoothing": 0.1}), ("ce_nomask", {})):
        logits = torch.from_numpy(z[tag + "_pred"]).requires_grad_(True)
        target = torch.from_numpy(z[tag + "_target"])
        mask = torch.from_numpy(z[tag + "_mask"].astype(np.int64)) if (tag + "_mask") in z else None
        loss = O.masked_ce(logits, target, mask=mask, patch_size=16, stride=4, **kw)
        assert abs(float(loss) - float(z[tag + "_loss"])) <= FP32_TOL * abs(float(z[tag + "_loss"]))
        loss.backward()
        assert rel_err(logits.grad, z[tag + "_dpred"]) < FP32_TOL


def test_block_and_cross_attention():
    z = G.load("blocks.npz")
    sd = G.perturbed_block_state("blk", 64, 256, seed=41)
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(z["block_x"]).requires_grad_(True)
    y = O.block(x, sd, "blk", 4)
    assert rel_err(y, z["block_y"]) < FP32_TOL
    y.backward(torch.from_numpy(z["block_dy"]))
    assert rel_err(x.grad, z["block_dx"]) < FP32_TOL
    assert rel_err(sd["blk.attn.qkv.weight"].grad, z["block_dqkv_w"]) < FP32_TOL
    assert rel_err(sd["blk.mlp.fc1.bias"].grad, z["block_dfc1_b"]) < FP32_TOL
    assert rel_err(sd["blk.norm1.weight"].grad, z["block_dnorm1_w"]) < FP32_TOL

    sdx = {k: v.requires_grad_(True) for k, v in G.perturbed_xattn_state("xa", 32, seed=43).items()}
    q = torch.from_numpy(z["xa_q"]).requires_grad_(True)
    c = torch.from_numpy(z["xa_c"]).requires_grad_(True)
    y = O.cross_attention(q, c, sdx, "xa", 4)
    assert rel_err(y, z["xa_y"]) < FP32_TOL
    y.backward(torch.from_numpy(z["xa_dy"]))
    assert rel_err(q.grad, z["xa_dq"]) < FP32_TOL
    assert rel_err(c.grad, z["xa_dc"]) < FP32_TOL
    assert rel_err(sdx["xa.kv.weight"].grad, z["xa_dkv_w"]) < FP32_TOL


def test_depth_standardisation():
    z = G.load("depth_std.npz")
    out = O.standardize_depth(torch.from_numpy(z["depth"]))
    assert rel_err(out, z["out"]) < 1e-6


def test_multivit_encoder_pass():
    z = G.load("small_multivit.npz")
    cfg = O.small_cfg()
    sd = {k: v.clone().requires_grad_(not k.endswith(O.FROZEN_SUFFIX)) for k, v in G.perturbed_state(cfg, seed=7).items()}
    x = G.seeded_inputs(cfg, B=2, seed=3)
    layers = O.multivit_forward(sd, cfg, x, return_all_layers=True)
    for i, t in enumerate(layers):
        assert rel_err(t.detach(), z[f"layer{i}"]) < FP32_TOL, i
    layers[-1].backward(torch.from_numpy(z["dy"]))
    for n in ("global_tokens", "input_adapters.depth.proj.weight", "encoder.0.attn.qkv.weight", "encoder.1.mlp.fc2.bias"):
        assert rel_err(sd[n].grad, z["grad_" + n]) < FP32_TOL, n
    assert rel_err(O.multivit_forward(sd, cfg, {"rgb": x["rgb"]}).detach(), z["rgb_only"]) < FP32_TOL
