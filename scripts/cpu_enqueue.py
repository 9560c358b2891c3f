"""Host time to enqueue one step against the device time of the step: python3 scripts/cpu_enqueue.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd.builders import build_losses, build_multimae
from multimae_amd.train import PretrainStep

if os.environ.get("MMAE_DP_SINGLE_RANK") == "1":  # the same through a one-rank RCCL group (bucket hooks, all-reduce calls)
    from multimae_amd.train import init_distributed_mode
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29537")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    init_distributed_mode()
torch.manual_seed(0)
model = build_multimae().cuda()
step = PretrainStep(model, build_losses(), num_encoded_tokens=98, alphas=1.0, lr=1e-4, weight_decay=0.05)
B = 256
x = {"rgb": torch.randn(B, 3, 224, 224, device="cuda"), "depth": torch.rand(B, 1, 224, 224, device="cuda")}
for _ in range(5):
    step(x)
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(20):
    t0 = time.perf_counter()
    step(x)
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / 20
host.sort()
print(f"host enqueue per step: median {host[10]*1e3:.2f} ms, min {host[0]*1e3:.2f}, max {host[-1]*1e3:.2f}; wall per step {wall*1e3:.2f} ms")
# the same with the device drained before every step: pure host cost of a step
pure = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(x)
    pure.append(time.perf_counter() - t0)
pure.sort()
print(f"host cost of a step on an idle device: median {pure[5]*1e3:.2f} ms")
