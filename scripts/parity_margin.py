"""How close to the 2e-2 bar the step is, and what bf16 costs the reference itself: config 1 (ViT-B/16 RGB-D 224^2, 98
visible tokens) at 2 / 4 / 8 images, every parameter gradient as ||g - g_ref||_2 / ||g_ref||_2 per tensor against the
fp32 oracle, for (a) this package's HIP step and (b) the same oracle under torch.autocast(bfloat16) on the host, i.e. the
reference's own mixed-precision arithmetic.  Prints one table (committed as profiles/r3_parity_margin.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import multimae_oracle as O
from tests.golden import common as G
from tests.conftest import rel_err
from tests.test_gpu_model import _build, _hip_step


def worst(gerr, n=3):
    return sorted(gerr.items(), key=lambda kv: -kv[1])[:n]


def all_as_one(g, ref):
    num = sum(((g[n].double().cpu() - ref[n].double()) ** 2).sum() for n in ref).sqrt().item()
    return num / sum((ref[n].double() ** 2).sum() for n in ref).sqrt().item()


if __name__ == "__main__":
    cfg = O.vit_b_rgbd_cfg()
    sd = G.perturbed_state(cfg, seed=1)
    k = 98
    model = _build(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    print("per-tensor error = ||g - g_fp32||_2 / ||g_fp32||_2 (L2 norm over a tensor, not element-wise); bar: 2e-2")
    print(f"{'images':>6} | {'HIP worst tensor':>16} | {'HIP all grads':>13} | {'autocast worst':>14} | {'autocast all':>12} | "
          f"{'HIP on autocast-worst':>21} | worst tensors (HIP) / (autocast)")
    for B in (2, 4, 8):
        x = G.seeded_inputs(cfg, B=B, seed=2)
        p, noises, noise_all = O.draw_mask_randomness(B, [196, 196], 1.0, torch.Generator().manual_seed(4))
        t0 = time.time()
        o_losses, o_preds, _, o_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k)
        a_losses, a_preds, _, a_grads = O.step_grads(sd, cfg, x, p, noises, noise_all, k, autocast_dtype=torch.bfloat16)
        preds, masks, losses, grads = _hip_step(model, cfg, x, (p, noises, noise_all), k)
        hip = {n: rel_err(grads[n].cpu(), o_grads[n]) for n in o_grads}
        ac = {n: rel_err(a_grads[n], o_grads[n]) for n in o_grads}
        hw, aw = worst(hip), worst(ac)
        print(f"{B:6d} | {hw[0][1]:16.2e} | {all_as_one(grads, o_grads):13.2e} | {aw[0][1]:14.2e} | {all_as_one(a_grads, o_grads):12.2e} | "
              f"{hip[aw[0][0]]:21.2e} | {[(n, f'{e:.1e}') for n, e in hw]} / {[(n, f'{e:.1e}') for n, e in aw]}", flush=True)
        lp = {t: (abs(losses[t].item() - float(o_losses[t])) / abs(float(o_losses[t])), abs(float(a_losses[t]) - float(o_losses[t])) / abs(float(o_losses[t]))) for t in o_losses}
        pp = {t: (rel_err(preds[t].float().cpu(), o_preds[t]), rel_err(a_preds[t].float(), o_preds[t])) for t in o_preds}
        print(f"       losses (HIP, autocast): { {t: (f'{a:.1e}', f'{b:.1e}') for t, (a, b) in lp.items()} }  predictions: "
              f"{ {t: (f'{a:.1e}', f'{b:.1e}') for t, (a, b) in pp.items()} }  [{time.time() - t0:.0f} s]", flush=True)
