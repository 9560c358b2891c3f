"""Times the GEMM / attention kernels at the shapes of the ViT-B RGB-D step (HIP events, random data)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import _lib
if os.environ.get("BENCH_LIB"):  # another build of the library (A/B on one box)
    _lib.LIB_PATH = os.path.abspath(os.environ["BENCH_LIB"])
from multimae_amd import ops


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def r(*s, dt=torch.bfloat16):
    return (torch.randn(*s, device="cuda") * 0.5).to(dt)


if __name__ == "__main__":
    M = 25344
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    shapes_nt = [] if only in ("tn", "attn") else [("qkv", M, 2304, 768, 0), ("proj+res", M, 768, 768, 2), ("fc1+gelu", M, 3072, 768, 1), ("fc2+res", M, 768, 3072, 2),
                 ("dgelu", M, 3072, 768, 3), ("dLN fc1", M, 768, 3072, 0), ("dA qkv", M, 768, 2304, 0), ("dA proj", M, 768, 768, 0), ("dec fc1", 50176, 1024, 256, 1), ("dec qkv", 50176, 768, 256, 0),
                 ("dec fc2", 50176, 256, 1024, 2), ("pctx", M, 256, 768, 0), ("dec dgelu", 50176, 1024, 256, 3), ("dec dLN fc1", 50176, 256, 1024, 0),
                 ("dec q", 50176, 256, 256, 0), ("dec kv", M, 512, 256, 0), ("dec proj f32", 50176, 256, 256, 4), ("dec dA qkv", 50176, 256, 768, 0),
                 ("L fc1+gelu", 18496, 4096, 1024, 1), ("L dgelu", 18496, 4096, 1024, 3), ("L qkv", 18496, 3072, 1024, 0)]
    if only == "dec":
        shapes_nt = [t for t in shapes_nt if t[0].startswith("dec") or t[0].startswith("L ")]
    if only == "cold":  # operands rotated through more than the 256-MB Infinity Cache: what a launch inside the step sees
        for name, m, n, k, epi in shapes_nt:
            if not (name.startswith("dec") or name in ("pctx", "qkv", "dA proj", "proj+res")):
                continue
            per = (m * k + (m * n) * (2 if epi in (2, 4) else 1) * (2 if epi == 1 else 1) + (m * n * (2 if epi == 2 else 1) if epi in (2, 3) else 0)) * 2
            copies = max(2, int(600e6 // per) + 1)
            As = [r(m, k) for _ in range(copies)]
            w = r(n, k)
            bias = torch.randn(n, device="cuda")
            auxs = [(r(m, n, dt=torch.float32) if epi == 2 else (r(m, n) if epi == 3 else None)) for _ in range(copies)]
            outs = [torch.empty(m, n, device="cuda", dtype=torch.float32 if epi in (2, 4) else torch.bfloat16) for _ in range(copies)]
            outs2 = [torch.empty(m, n, device="cuda", dtype=torch.bfloat16) if epi == 1 else None for _ in range(copies)]
            res = []
            for variant in (1, 5, 7, 0):
                ops._L().mmae_gemm_nt_set_variant(variant)
                it = [0]
                def call():
                    i = it[0] % copies; it[0] += 1
                    ops.gemm_nt(epi, As[i], w, None if epi == 3 else bias, aux=auxs[i], out=outs[i])
                t = timeit(call, iters=4 * copies)
                res.append(f"{t*1e6:7.1f} us")
            ops._L().mmae_gemm_nt_set_variant(0)
            print(f"cold {name:12s} M={m} N={n} K={k} ({copies} operand sets): t128 {res[0]} | pp256 {res[1]} | 2wg128x256 {res[2]} | auto {res[3]}")
        sys.exit(0)
    if only == "pp":  # the 256x256 kernel alone, at the shapes it is taken for
        for name, m, n, k, epi in shapes_nt:
            if name.startswith("dec") or name == "pctx":
                continue
            a, w = r(m, k), r(n, k)
            bias = torch.randn(n, device="cuda")
            aux = r(m, n, dt=torch.float32) if epi == 2 else (r(m, n) if epi == 3 else None)
            ops._L().mmae_gemm_nt_set_variant(5)
            res = []
            for tails in (0, 2, 1):  # no tail tiles, 128-row tail tiles, 128- or 64-row by cost
                ops._L().mmae_gemm_nt_set_half_tiles(tails)
                t = timeit(lambda: ops.gemm_nt(epi, a, w, None if epi == 3 else bias, aux=aux), iters=30)
                res.append(f"{t*1e6:8.1f} us {2*m*n*k/t/1e12:7.1f} TF")
            ops._L().mmae_gemm_nt_set_half_tiles(1)
            print(f"pp256 {name:12s} M={m} N={n} K={k}: full tiles only {res[0]} | 128-row tail {res[1]} | by cost {res[2]} (tail rows {ops._L().mmae_gemm_nt_last_tail_rows()})")
        sys.exit(0)
    for name, m, n, k, epi in shapes_nt:
        a, w = r(m, k), r(n, k)
        bias = torch.randn(n, device="cuda")
        aux = r(m, n, dt=torch.float32) if epi == 2 else (r(m, n) if epi == 3 else None)
        res = []
        for variant in (1, 4, 6, 5, 7, 0):  # 6 = ping-pong kernel with full tiles only, 7 = 128x256 tiles, two workgroups per CU
            ops._L().mmae_gemm_nt_set_variant(5 if variant == 6 else variant)
            ops._L().mmae_gemm_nt_set_half_tiles(0 if variant == 6 else 1)
            t = timeit(lambda: ops.gemm_nt(epi, a, w, None if epi == 3 else bias, aux=aux))
            res.append(f"{t*1e6:8.1f} us {2*m*n*k/t/1e12:7.1f} TF")
        ops._L().mmae_gemm_nt_set_variant(0)
        # yardstick only (not a product path): the vendor library's plain bf16 product of the same shape, no epilogue
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: torch.matmul(a, w.t(), out=out))
        print(f"gemm_nt {name:10s} M={m} N={n} K={k}: t128 {res[0]} | p128 {res[1]} | pp256 full {res[2]} | pp256 {res[3]} | 2wg128x256 {res[4]} | auto {res[5]}"
              f" | vendor plain {t*1e6:8.1f} us {2*m*n*k/t/1e12:7.1f} TF")
    if len(sys.argv) > 1 and sys.argv[1] in ("nt", "dec"):
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tn":
        pass
    for name, m, n, k in [] if only == "attn" else [("dW qkv", M, 2304, 768), ("dW proj", M, 768, 768), ("dW fc1", M, 3072, 768), ("dW fc2", M, 768, 3072),
                          ("dW dec fc1", 50176, 1024, 256), ("dW dec proj", 50176, 256, 256)]:
        x, y = r(m, n), r(m, k)
        out = torch.zeros(n, k, device="cuda"); cs = torch.zeros(n, device="cuda")
        res = []
        for variant in (1, 2):
            ops._L().mmae_gemm_tn_set_variant(variant)
            t = timeit(lambda: ops.gemm_tn(x, y, out=out, accumulate=True, colsum_out=cs, colsum_accumulate=True))
            res.append(f"{t*1e6:8.1f} us {2*m*n*k/t/1e12:7.1f} TF")
        ops._L().mmae_gemm_tn_set_variant(2)
        t = timeit(lambda: ops.gemm_tn(x, y, out=out, accumulate=True))
        res.append(f"{t*1e6:8.1f} us")
        ops._L().mmae_gemm_tn_set_phase(1)
        t = timeit(lambda: ops.gemm_tn(x, y, out=out, accumulate=True, colsum_out=cs, colsum_accumulate=True))
        res.append(f"{t*1e6:8.1f} us")
        ops._L().mmae_gemm_tn_set_phase(0)
        ops._L().mmae_gemm_tn_set_variant(0)
        o2 = torch.empty(n, k, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: torch.matmul(x.t(), y, out=o2))  # yardstick: vendor library, bf16 output, no column sums
        print(f"gemm_tn {name:10s} M={m} N={n} K={k}: t128 {res[0]} | pp256 {res[1]} | pp256 without column sums {res[2]} | pp256 kernel alone {res[3]} | vendor plain {t*1e6:8.1f} us {2*m*n*k/t/1e12:7.1f} TF")
    for name, B, H, nq, nk, d in [("enc", 256, 12, 99, 99, 64), ("dec self", 256, 8, 196, 196, 32), ("dec cross", 256, 8, 196, 99, 32)]:
        if nq == nk:
            qkv = r(B * nq, 3 * H * d); q, k_, v = qkv[:, :H*d], qkv[:, H*d:2*H*d], qkv[:, 2*H*d:]
            dqkv = torch.empty_like(qkv); dq, dk, dv = dqkv[:, :H*d], dqkv[:, H*d:2*H*d], dqkv[:, 2*H*d:]
        else:
            q = r(B * nq, H * d); kv = r(B * nk, 2 * H * d); k_, v = kv[:, :H*d], kv[:, H*d:]
            dq = torch.empty_like(q); dkv = torch.empty_like(kv); dk, dv = dkv[:, :H*d], dkv[:, H*d:]
        o, lse = ops.attn_fwd(q, k_, v, B, H, nq, nk, d, d ** -0.5)
        do = r(B * nq, H * d)
        t = timeit(lambda: ops.attn_fwd(q, k_, v, B, H, nq, nk, d, d ** -0.5))
        fl = 4.0 * B * H * nq * nk * d
        bw = []
        for variant in (1, 2):
            ops._L().mmae_attn_bwd_set_variant(variant)
            t2 = timeit(lambda: ops.attn_bwd(q, k_, v, o, do, lse, dq, dk, dv, B, H, nq, nk, d, d ** -0.5))
            bw.append(f"{t2*1e6:7.1f} us {2*fl/t2/1e12:6.1f} TF")
        ops._L().mmae_attn_bwd_set_variant(0)
        print(f"attn {name:10s}: fwd {t*1e6:7.1f} us {fl/t/1e12:6.1f} TF | bwd two kernels {bw[0]} | fused {bw[1]}")
