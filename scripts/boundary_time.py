"""Device time between the end of one step's backward and the first encoder block of the next (gradient norm, optimizer tail,
gradient clear, depth standardisation, mask generation, patch packing and projection), from events on the main stream of
the un-profiled step: python3 scripts/boundary_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import functional as Fn
from multimae_amd.builders import build_losses, build_multimae
from multimae_amd.train import PretrainStep

B = 256
x = {"rgb": torch.randn(B, 3, 224, 224, device="cuda"), "depth": torch.rand(B, 1, 224, 224, device="cuda")}
torch.manual_seed(0)
model = build_multimae().cuda()
step = PretrainStep(model, build_losses(), num_encoded_tokens=98, alphas=1.0, lr=1e-4, weight_decay=0.05, standardize_depth=True)
marks = []
orig_blocks, orig_gn = Fn.run_blocks, step.opt.grad_norm_sq


def blocks(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("blocks", e))
    return orig_blocks(*a, **k)


def gn():
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("backward_done", e))
    return orig_gn()


orig_join = Fn.join_side_stream


def join(*a, **k):
    r = orig_join(*a, **k)
    if step._armed is False and marks and marks[-1][0] == "blocks":  # the join after backward, once per step
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("backward_done", e))
    return r


Fn.join_side_stream = join
import multimae_amd.train as TR
TR.Fn.join_side_stream = join
Fn.run_blocks = blocks
import multimae_amd.multimae as MM
MM.orig_join = Fn.join_side_stream


def join(*a, **k):
    r = orig_join(*a, **k)
    if step._armed is False and marks and marks[-1][0] == "blocks":  # the join after backward, once per step
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("backward_done", e))
    return r


Fn.join_side_stream = join
import multimae_amd.train as TR
TR.Fn.join_side_stream = join
Fn.run_blocks = blocks
for _ in range(25):
    step(x)
torch.cuda.synchronize()
ev = marks[2 * 5:]
bound = [ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(1, len(ev) - 1, 2)]   # backward_done -> next blocks
body = [ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(0, len(ev) - 1, 2)]    # blocks -> backward_done
print("order:", [n for n, _ in ev[:4]])
print(f"step boundary (backward done -> first encoder block): median {sorted(bound)[len(bound)//2]:.3f} ms, min {min(bound):.3f}, max {max(bound):.3f}")
print(f"encoder + decoders + losses + backward: median {sorted(body)[len(body)//2]:.3f} ms")
