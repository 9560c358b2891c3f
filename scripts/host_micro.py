"""Host cost of the stream primitives used per launch: python3 scripts/host_micro.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
side = torch.cuda.Stream()
x = torch.empty(1024, device="cuda")
def t(fn, n=2000):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
def ctx():
    with torch.cuda.stream(side):
        pass
print("with torch.cuda.stream(side): %.1f us" % t(ctx))
print("side.wait_stream(current):    %.1f us" % t(lambda: side.wait_stream(torch.cuda.current_stream())))
print("x.record_stream(side):        %.1f us" % t(lambda: x.record_stream(side)))
print("torch.cuda.current_stream():  %.1f us" % t(lambda: torch.cuda.current_stream()))
print("torch.empty(1024):            %.1f us" % t(lambda: torch.empty(1024, device="cuda")))
ev = torch.cuda.Event()
print("event record + wait:          %.1f us" % t(lambda: (ev.record(), side.wait_event(ev))))
torch.cuda.synchronize()
