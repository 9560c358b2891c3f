"""Live check of the oracle against the imported reference at full ViT-B size.
Runs only where the reference tree is mounted (the build container); skipped elsewhere."""
import os
import sys
import warnings

import pytest
import torch

from oracle import multimae_oracle as O
from tests.conftest import rel_err
from tests.golden import common as G

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "multimae")), reason="reference tree not mounted")


def _ref_model(cfg):
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    warnings.filterwarnings("ignore")
    from tests.golden.make_golden import build_ref_model
    return build_ref_model(cfg)


def test_vit_b_rgbd_step_matches_reference():
    cfg = O.vit_b_rgbd_cfg()
    sd = G.perturbed_state(cfg, seed=1)
    model = _ref_model(cfg)
    import multimae as ref
    model.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 94_006_016
    B, k = 2, 98
    x = G.seeded_inputs(cfg, B=B, seed=2)
    from tests.golden.make_golden import replay_mask_rng
    p, noises, noise_all = replay_mask_rng(9, B, [196, 196], 1.0, False)
    torch.manual_seed(9)
    preds, masks = model(x, num_encoded_tokens=k, alphas=1.0)
    fns = {"rgb": ref.MaskedMSELoss(16, 1), "depth": ref.MaskedL1Loss(16, 1),
           "norm_rgb": ref.MaskedMSELoss(16, 1, norm_pix=True)}
    tgt = {"rgb": "rgb", "depth": "depth", "norm_rgb": "rgb"}
    losses = {t: fns[t](preds[t].float(), x[tgt[t]], mask=masks[tgt[t]]) for t in preds}
    sum(losses.values()).backward()

    o_losses, o_preds, o_masks, o_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k)
    for d in masks:
        assert torch.equal(masks[d], o_masks[d])
    for t in preds:
        assert rel_err(o_preds[t], preds[t].detach()) < 1e-4, t
        assert abs(float(o_losses[t]) - float(losses[t])) / abs(float(losses[t])) < 1e-4, t
    worst = 0.0
    for name, prm in model.named_parameters():
        if prm.requires_grad:
            assert prm.grad is not None, name
            e = rel_err(o_grads[name], prm.grad)
            worst = max(worst, e)
            assert e < 1e-4, (name, e)
    print("worst grad rel err", worst)
