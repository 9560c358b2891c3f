"""bench.py against another build of the kernel library (A/B on one box), optionally with a stand-in for a collective
holding CUs part of the time:
  [HOG=blocks,busy_us,period_us] python scripts/hog/bench_with_lib.py <lib.so> [bench args]
HOG: a thread launches the hog kernel (scripts/hog/hog.hip) on a stream of its own, `blocks` workgroups for `busy_us`
every `period_us`."""
import ctypes, os, runpy, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from multimae_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
stop = threading.Event()
if os.environ.get("HOG"):
    import torch
    blocks, busy, period = (int(v) for v in os.environ["HOG"].split(","))
    hog = ctypes.CDLL(os.path.join(HERE, "libhog.so"))
    hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
    hog.hog_fat_launch.argtypes = hog.hog_launch.argtypes
    launch = hog.hog_fat_launch if os.environ.get("HOG_FAT", "0") == "1" else hog.hog_launch

    def loop():
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        while not stop.is_set():
            launch(blocks, 256, busy, st.cuda_stream)
            st.synchronize()  # at most one hog kernel queued
            time.sleep(max(period - busy, 0) * 1e-6)

    threading.Thread(target=loop, daemon=True).start()
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    stop.set()
