"""Host-side meters the reference's training loop wraps around the step (utils/logger.py:28-233 there: SmoothedValue,
MetricLogger, WandbLogger; used at run_pretraining_multimae.py:395,411-421,524-557).  No device work except the one
cross-rank sum in synchronize_between_processes (the reference's collective C4)."""
import datetime
import statistics
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist


class SmoothedValue:
    """Sliding window (median / avg / max / value) plus the running global average of a series."""

    def __init__(self, window_size=20, fmt=None):
        self.deque, self.total, self.count = deque(maxlen=window_size), 0.0, 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """Sums count and total over the ranks (not the window), as the reference does."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), float(t[1].item())

    median = property(lambda s: statistics.median_low(s.deque))  # torch.median's choice for an even count: the lower one
    avg = property(lambda s: sum(s.deque) / len(s.deque))
    global_avg = property(lambda s: s.total / s.count)
    max = property(lambda s: max(s.deque))
    value = property(lambda s: s.deque[-1])

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters, self.delimiter = defaultdict(SmoothedValue), delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is not None:
                self.meters[k].update(v.item() if isinstance(v, torch.Tensor) else v)

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def __getattr__(self, attr):
        meters = self.__dict__.get("meters", {})
        if attr in meters:
            return meters[attr]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'")

    def __str__(self):
        return self.delimiter.join(f"{k}: {m}" for k, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def log_every(self, iterable, print_freq, header=None):
        """Yields the items of `iterable`; every print_freq items prints progress, the meters, iteration / data time."""
        n, header = len(iterable), header or ""
        it_t, data_t = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        start = end = time.time()
        for i, obj in enumerate(iterable):
            data_t.update(time.time() - end)
            yield obj
            it_t.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = datetime.timedelta(seconds=int(it_t.global_avg * (n - i)))
                mem = f"\tmax mem: {torch.cuda.max_memory_allocated() / 2 ** 20:.0f}" if torch.cuda.is_available() else ""
                print(self.delimiter.join([header, f"[{i:{len(str(n))}d}/{n}]", f"eta: {eta}", str(self), f"time: {it_t}",
                                           f"data: {data_t}"]) + mem)
            end = time.time()
        total = time.time() - start
        print(f"{header} Total time: {datetime.timedelta(seconds=int(total))} ({total / max(n, 1):.4f} s / it)")


class WandbLogger:
    """The reference's wandb front end with no back end (there is no network service here): keeps the step counter and the
    last metrics so a loop written against it runs unchanged."""

    def __init__(self, args=None):
        self.step, self.last = 0, {}

    def set_step(self, step=None):
        self.step = step if step is not None else self.step + 1

    def update(self, metrics):
        self.last = {k: (v.item() if isinstance(v, torch.Tensor) else v) for k, v in metrics.items() if v is not None}

    def flush(self):
        pass

    def finish(self):
        pass
