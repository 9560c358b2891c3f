"""Device time of the phases of a step (events on the main stream): python3 scripts/phase_times.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import functional as Fn
from multimae_amd.builders import build_losses, build_multimae
from multimae_amd.train import PretrainStep

torch.manual_seed(0)
model = build_multimae().cuda()
step = PretrainStep(model, build_losses(), num_encoded_tokens=98, alphas=1.0, lr=1e-4, weight_decay=0.05)
B = 256
x = {"rgb": torch.randn(B, 3, 224, 224, device="cuda"), "depth": torch.rand(B, 1, 224, 224, device="cuda")}
for _ in range(5):
    step(x)
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
acc = [0.0] * 4
N = 10
for _ in range(N):
    e = [ev() for _ in range(5)]
    e[0].record()
    step.flat.zero_grad()
    preds, masks = step.model({d: t for d, t in x.items()}, num_encoded_tokens=98, alphas=1.0)
    e[1].record()
    task_losses = {t: fn(preds[t], x[tgt], mask=masks.get(tgt)) for t, (fn, tgt) in step.losses.items()}
    loss = sum(task_losses.values())
    e[2].record()
    loss.backward()
    step.buckets.finish()
    Fn.join_side_stream()
    e[3].record()
    gn2 = step.opt.grad_norm_sq()
    step.opt.step(grad_scale=1.0, grad_sumsq=gn2)
    e[4].record()
    torch.cuda.synchronize()
    for i in range(4):
        acc[i] += e[i].elapsed_time(e[i + 1])
print("forward %.2f ms | losses %.2f | backward (+ joins) %.2f | optimizer %.2f | sum %.2f" % (*[a / N for a in acc], sum(acc) / N))
