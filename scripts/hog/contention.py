"""GEMM time with some CUs held by a long-running kernel of another stream (scripts/hog/hog.hip; build it in place with
`hipcc --offload-arch=gfx950 -O3 -fPIC -shared scripts/hog/hog.hip -o scripts/hog/libhog.so`).
  python scripts/hog/contention.py [path of libmmae_hip.so to load]
Ten back-to-back launches of a GEMM on the main stream, timed by events, while `H` hog workgroups sit on the chip."""
import ctypes, os, sys, time
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from multimae_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from multimae_amd import ops

hog = ctypes.CDLL(os.path.join(HERE, "libhog.so"))
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
hog.hog_fat_launch.argtypes = hog.hog_launch.argtypes
FAT = os.environ.get("HOG_FAT", "0") == "1"  # the hog with a collective kernel's register / LDS footprint
side = torch.cuda.Stream()


def r(*s, dt=torch.bfloat16):
    return (torch.randn(*s, device="cuda") * 0.5).to(dt)


def run(fn, H, n=10, hog_us=6000):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        if H:
            (hog.hog_fat_launch if FAT else hog.hog_launch)(H, 256, hog_us, side.cuda_stream)
            time.sleep(0.0005)  # let the hog start
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


M = 25344
print("library:", _lib.LIB_PATH)
for name, m, n_, k, epi in [("qkv", M, 2304, 768, 0), ("fc1+gelu", M, 3072, 768, 1), ("dA fc1", M, 768, 3072, 0), ("dec fc1", 50176, 1024, 256, 1)]:
    a, w, bias = r(m, k), r(n_, k), torch.randn(n_, device="cuda")
    fn = lambda: ops.gemm_nt(epi, a, w, bias)
    fn(); torch.cuda.synchronize()
    line = f"{name:9s}"
    for H in (0, 8, 32, 64):
        line += f" | hog {H:2d}: {run(fn, H):7.1f} us"
    print(line, flush=True)
