"""Step time against the number of decoders (same encoder): python3 scripts/decoder_cost.py
The difference between the 3-decoder and the 1-decoder step, halved, is what one decoder (forward, loss, backward, its share
of the optimizer) costs inside the step; printed with the streams on (as measured by bench.py) and off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import functional as Fn
from multimae_amd.builders import build_losses, build_multimae
from multimae_amd.train import PretrainStep

B = 256
x = {"rgb": torch.randn(B, 3, 224, 224, device="cuda"), "depth": torch.rand(B, 1, 224, 224, device="cuda")}


def run(out_domains, norm_pix, streams):
    Fn.SIDE_STREAM_DW = Fn.DECODER_STREAMS = streams
    torch.manual_seed(0)
    model = build_multimae(out_domains=out_domains, extra_norm_pix_loss=norm_pix).cuda()
    step = PretrainStep(model, build_losses(out_domains, extra_norm_pix_loss=norm_pix), num_encoded_tokens=98, alphas=1.0, lr=1e-4, weight_decay=0.05)
    for _ in range(4):
        step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    del step, model
    torch.cuda.empty_cache()
    return ms


for streams in (True, False):
    one = run(("depth",), False, streams)
    two = run(("rgb", "depth"), False, streams)
    three = run(("rgb", "depth"), True, streams)
    print(f"streams {'on ' if streams else 'off'}: 1 decoder {one:.2f} ms, 2 decoders {two:.2f} ms, 3 decoders {three:.2f} ms "
          f"-> per decoder {(three - one) / 2:.2f} ms, encoder + input side + optimizer about {one - (three - one) / 2:.2f} ms", flush=True)
