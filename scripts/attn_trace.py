"""Timing build only (libmmae_trace.so): lifetime of the workgroups of the fused attention backward kernel
(start -> operands staged -> this wave's items done), median over the first 1,024 workgroups."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multimae_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "libmmae_trace.so")
from multimae_amd import ops
raw = ctypes.CDLL(_lib.LIB_PATH)


def r(*s):
    return (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)


for name, B, H, nq, nk, d in [("enc", 256, 12, 99, 99, 64), ("dec self", 256, 8, 196, 196, 32), ("dec cross", 256, 8, 196, 99, 32)]:
    if nq == nk:
        qkv = r(B * nq, 3 * H * d); q, k_, v = qkv[:, :H*d], qkv[:, H*d:2*H*d], qkv[:, 2*H*d:]
        dqkv = torch.empty_like(qkv); dq, dk, dv = dqkv[:, :H*d], dqkv[:, H*d:2*H*d], dqkv[:, 2*H*d:]
    else:
        q = r(B * nq, H * d); kv = r(B * nk, 2 * H * d); k_, v = kv[:, :H*d], kv[:, H*d:]
        dq = torch.empty_like(q); dkv = torch.empty_like(kv); dk, dv = dkv[:, :H*d], dkv[:, H*d:]
    o, lse = ops.attn_fwd(q, k_, v, B, H, nq, nk, d, d ** -0.5)
    do = r(B * nq, H * d)
    ops._L().mmae_attn_bwd_set_variant(2)
    for _ in range(3):
        ops.attn_bwd(q, k_, v, o, do, lse, dq, dk, dv, B, H, nq, nk, d, d ** -0.5)
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 8 * 8, dtype=np.uint64)
    assert raw.mmae_debug_attn_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(1024, 8, 8).astype(np.float64) * 0.01
    t0 = t[:, :, 0].min()
    stage = t[:, :, 1] - t[:, :, 0]
    items = t[:, :, 2] - t[:, :, 1]
    life = t[:, :, 2].max(axis=1) - t[:, :, 0].min(axis=1)
    print(f"{name}: staging {np.median(stage):.2f} us (p90 {np.percentile(stage, 90):.2f}); items per wave median {np.median(items):.2f} us, "
          f"slowest wave of a workgroup {np.median(items.max(axis=1)):.2f}, fastest {np.median(items.min(axis=1)):.2f}; workgroup lifetime {np.median(life):.2f} us; "
          f"first 1,024 workgroups span {t[:, :, 2].max() - t0:.1f} us")
    print(f"   staging: loads issued after {np.median(t[:, :, 3] - t[:, :, 0]):.2f} us, landed and stored after {np.median(t[:, :, 4] - t[:, :, 0]):.2f}, "
          f"past the first barrier after {np.median(t[:, :, 5] - t[:, :, 0]):.2f}, staged after {np.median(stage):.2f}")
    print("   per wave (median items time):", " ".join(f"{np.median(items[:, w]):.2f}" for w in range(8)))
