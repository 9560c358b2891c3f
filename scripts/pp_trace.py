"""Timing build only (libmmae_trace.so, -DMMAE_PP_TRACE): per-tile timestamps of the 256x256 kernel.
For each epilogue kind: main-loop time per tile, time from the end of the main loop to the last store issued, and from
there to the end of the next tile's first k-tile (the wait that covers the stores)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multimae_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "libmmae_trace.so")
from multimae_amd import ops

L = ops._L()
raw = ctypes.CDLL(_lib.LIB_PATH)


def r(*s, dt=torch.bfloat16):
    return (torch.randn(*s, device="cuda") * 0.5).to(dt)


for name, m, n, k, epi in [("qkv", 25344, 2304, 768, 0), ("fc1+gelu", 25344, 3072, 768, 1), ("dgelu", 25344, 3072, 768, 3),
                           ("fc2+res", 25344, 768, 3072, 2), ("dA qkv", 25344, 768, 2304, 0)]:
    a, w = r(m, k), r(n, k)
    bias = torch.randn(n, device="cuda")
    aux = r(m, n, dt=torch.float32) if epi == 2 else (r(m, n) if epi == 3 else None)
    L.mmae_gemm_nt_set_variant(5)
    L.mmae_gemm_nt_set_half_tiles(0)
    for _ in range(3):
        ops.gemm_nt(epi, a, w, None if epi == 3 else bias, aux=aux)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 2 * 128, dtype=np.uint64)
    assert raw.mmae_debug_pp_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(256, 2, 16, 8).astype(np.float64) * 0.01  # us
    tiles = (m + 255) // 256 * ((n + 255) // 256)
    full = tiles // 256  # tiles every workgroup has
    t0 = t[:, :, 0, 7].min()
    print(f"{name}: {tiles} tiles, {tiles/256:.2f} per CU; kernel span {t.max() - t0:.1f} us")
    med = lambda x: float(np.median(x))
    for g in (0, 1):
        print(f"  group {g}: first k-tile done {med(t[:, g, 0, 5] - t[:, g, 0, 7]):.2f} us after the start")
        for i in range(1, full + 1):
            rec, prev = t[:, g, i], t[:, g, i - 1]
            line = f"    tile {i}: rest of main loop {med(rec[:, 0] - prev[:, 5]):6.2f} | epilogue issue {med(rec[:, 1] - rec[:, 0]):5.2f}"
            if i < full:
                line += (f" | to first barrier {med(rec[:, 2] - rec[:, 1]):5.2f} | to phase-3 wait {med(rec[:, 3] - rec[:, 2]):5.2f} | in the wait {med(rec[:, 4] - rec[:, 3]):5.2f}"
                         f" | to end of k-tile {med(rec[:, 5] - rec[:, 4]):5.2f} | period {med(rec[:, 5] - prev[:, 5]):6.2f}")
            print(line)
