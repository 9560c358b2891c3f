"""Times LayerNorm forward / backward at the step's two shapes, cold (the buffer set rotates past the 256 MiB last-level
cache), for both backward kernels.  Usage: python scripts/bench_ln.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimae_amd import ops


def timeit(fns, iters=40):
    for f in fns: f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): fns[i % len(fns)]()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    torch.manual_seed(0)
    for rows, dim in ((25344, 768), (50176, 256), (25344, 256)):
        sets = []
        for _ in range(4):
            x = torch.randn(rows, dim, device="cuda"); g = torch.randn(dim, device="cuda"); b = torch.randn(dim, device="cuda")
            y, mean, rstd = ops.layernorm_fwd(x, g, b)
            dy = torch.randn(rows, dim, device="cuda").bfloat16(); dres = torch.randn(rows, dim, device="cuda")
            gg = torch.zeros(dim, device="cuda"); gb = torch.zeros(dim, device="cuda")
            sets.append((x, g, b, mean, rstd, dy, dres, gg, gb))
        t = timeit([lambda s=s: ops.layernorm_fwd(s[0], s[1], s[2]) for s in sets])
        line = f"ln rows={rows} dim={dim}: fwd {t*1e6:6.1f} us {rows*dim*6/t/1e12:5.2f} TB/s"
        for variant in (1, 0):
            ops._L().mmae_layernorm_bwd_set_variant(variant)
            for shadow, nbytes in ((True, 12), (False, 10)):
                t = timeit([lambda s=s: ops.layernorm_bwd(s[5], s[0], s[1], s[3], s[4], dres=s[6], want_bf16=shadow, defer_params=True) for s in sets])
                line += f" | bwd v{variant}{'+bf16' if shadow else ''} {t*1e6:6.1f} us {rows*dim*nbytes/t/1e12:5.2f} TB/s"
        ops._L().mmae_layernorm_bwd_set_variant(0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
