"""cProfile of the host side of one step (device drained before each step): python3 scripts/host_profile.py"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
pr = cProfile.Profile()
for _ in range(10):
    torch.cuda.synchronize()
    pr.enable()
    step(x)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
