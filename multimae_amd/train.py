"""Step-side runtime: flat parameter / gradient storage, fused AdamW, and data-parallel gradient exchange.

Replaces, for the pretraining step, the reference's DistributedDataParallel wrap (run_pretraining_multimae.py:224-226),
its NCCL bring-up (utils/dist.py:66-108) and the optimizer step inside NativeScalerWithGradNormCount
(utils/native_scaler.py:23-51).  bf16 needs no loss scaling, so there is no GradScaler.

Layout: every trainable parameter is a view into one fp32 buffer, ordered [decayed | not decayed], each region in
reverse registration order (the order in which backward produces gradients), so that a bucket of the gradient
buffer is complete as soon as backward has passed its last parameter.  Gradients are views into a second buffer of
the same layout; buckets are all-reduced (SUM) over RCCL as they complete, overlapped with the rest of backward,
and the 1/world factor is folded into the optimizer kernel."""
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from . import functional as Fn
from . import ops


def init_from_env(backend: str = "nccl"):
    """Environment-variable rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), one process per GPU.
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    _start_group(backend, rank, world, local)
    return rank, world, local


def _start_group(backend, rank, world, gpu, init_method=None, always=False):
    single = os.environ.get("MMAE_DP_SINGLE_RANK") == "1"  # rehearsal: a one-rank group still runs every collective
    if (world > 1 or single or always) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        native = os.environ.get("MMAE_COMM") == "native"
        kw = {} if init_method in (None, "", "env://") else {"init_method": init_method}
        if backend == "nccl":
            torch.cuda.set_device(gpu)
            # the step's own streams are chosen by timing idle kernels on candidates (functional.concurrent_stream): do it
            # before the communicator exists, so the timing never runs beside a resident collective kernel
            Fn.prepare_streams(torch.device("cuda", gpu), 2)
        if backend == "nccl" and not native:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu), **kw)
        else:
            # With the C ABI's RCCL driver (MMAE_COMM=native) torch.distributed is the host-side control plane only (the
            # communicator id, barriers, scalars): gloo.  The driver's communicator then is the only RCCL communicator of
            # the process (NativeComm refuses to be built beside a torch "nccl" group on the same GPU).
            dist.init_process_group(backend="gloo" if backend == "nccl" else backend, rank=rank, world_size=world, **kw)
        dist.barrier()


def init_distributed_mode(args="nccl"):
    """The reference's call shape (utils/dist.py:66-108 there; called as utils.init_distributed_mode(args) at
    run_pretraining_multimae.py:144): reads args.dist_on_itp / args.gpus / args.dist_url and the launcher's environment
    (OpenMPI, torchrun's RANK / WORLD_SIZE / LOCAL_RANK, or SLURM_PROCID), sets args.rank / world_size / gpu / distributed /
    dist_backend, starts the RCCL group ("nccl" is RCCL on ROCm) and returns None; without a launcher it sets
    args.distributed = False.  Given a backend string instead it is init_from_env(backend) and returns that tuple."""
    if isinstance(args, str):
        return init_from_env(args)
    env = os.environ
    gpus = getattr(args, "gpus", None)

    def pick(i):  # args.gpus[local rank]; the argument parser's placeholder default ([-1]) means "the local rank itself"
        g = gpus[i] if isinstance(gpus, (list, tuple)) and i < len(gpus) else None
        return i if g is None or g < 0 else g
    if getattr(args, "dist_on_itp", False):
        args.rank, args.world_size = int(env["OMPI_COMM_WORLD_RANK"]), int(env["OMPI_COMM_WORLD_SIZE"])
        args.gpu = pick(int(env["OMPI_COMM_WORLD_LOCAL_RANK"]))
        args.dist_url = "tcp://%s:%s" % (env["MASTER_ADDR"], env["MASTER_PORT"])
        env["LOCAL_RANK"], env["RANK"], env["WORLD_SIZE"] = str(args.gpu), str(args.rank), str(args.world_size)
    elif "RANK" in env and "WORLD_SIZE" in env:
        args.rank, args.world_size = int(env["RANK"]), int(env["WORLD_SIZE"])
        args.gpu = pick(int(env["LOCAL_RANK"]))
    elif "SLURM_PROCID" in env:
        args.rank = int(env["SLURM_PROCID"])
        args.world_size = int(env.get("SLURM_NTASKS", getattr(args, "world_size", 1) or 1))
        args.gpu = pick(args.rank % max(torch.cuda.device_count(), 1))
    else:
        print("Not using distributed mode")
        args.distributed = False
        return None
    args.distributed = True
    args.dist_backend = getattr(args, "dist_backend", None) or "nccl"
    if args.dist_backend == "nccl":
        torch.cuda.set_device(args.gpu)
    print("| distributed init (rank {}): {}, gpu {}".format(args.rank, getattr(args, "dist_url", "env://"), args.gpu), flush=True)
    # the reference starts a group for any world size, one rank included
    _start_group(args.dist_backend, args.rank, args.world_size, args.gpu, getattr(args, "dist_url", None), always=True)
    return None


def split_decay(model: torch.nn.Module):
    """(decayed, not decayed) trainable parameters: no decay for 1-D tensors and the names in model.no_weight_decay()
    (the usual rule of the reference family's optimizer factory, which is not part of the reference tree)."""
    skip = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim <= 1 or name.endswith(".bias") or any(name == s or name.startswith(s + ".") for s in skip):
            no_decay.append((name, p))
        else:
            decay.append((name, p))
    return decay, no_decay


class FlatParams:
    """Moves the trainable parameters of `model` into one flat fp32 buffer (and their .grad into another), laid out in
    the order backward completes their gradients so that contiguous buckets fill front to back."""

    def __init__(self, model: torch.nn.Module):
        decay, no_decay = split_decay(model)

        def backward_order(named):
            # reverse registration order within a module group; groups in the order backward reaches them: the output
            # adapters (registered before the encoder in MultiMAE, run after it), the encoder, the input side
            rank = lambda name: 0 if name.startswith("output_adapters.") else (2 if name.startswith("input_adapters.") or "global_tokens" in name else 1)
            return sorted(named[::-1], key=lambda np_: rank(np_[0]))
        decay, no_decay = backward_order(decay), backward_order(no_decay)
        self.named = decay + no_decay
        # frozen parameters and buffers: not in flat storage, but part of what a data-parallel start-up broadcasts
        self.frozen = [p for _, p in model.named_parameters() if not p.requires_grad] + [b for _, b in model.named_buffers()]
        dev = self.named[0][1].device

        def padded(n):  # keep every view 16-byte aligned, in the fp32 buffers and in their bf16 copies
            return (n + 7) // 8 * 8
        self.offsets, off = [], 0
        for _, p in decay:
            self.offsets.append(off)
            off += padded(p.numel())
        self.n_decay = off
        for _, p in no_decay:
            self.offsets.append(off)
            off += padded(p.numel())
        self.numel = off
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for (name, p), o in zip(self.named, self.offsets):
                view = self.params[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grads[o:o + p.numel()].view(p.shape)
                p._mmae_direct = True  # kernels may accumulate this parameter's gradient in place
        Fn.clear_weight_cache()
        # bf16 copies for the GEMMs: W ([out,in], same layout as the fp32 master, written by the AdamW kernel) and W^T
        # (one multi-tensor transpose launch); every 2-D Linear weight views into these two flat buffers
        self.w_bf16 = self.wt_bf16 = self._tr_table = None
        if dev.type == "cuda":
            self.w_bf16 = torch.empty(off, dtype=torch.bfloat16, device=dev)
            self.wt_bf16 = torch.empty_like(self.w_bf16)
            rows, tiles = [], 0
            for (name, p), o in zip(self.named, self.offsets):
                if p.ndim == 2:
                    n, k = p.shape
                    rows.append([o, n, k, tiles])
                    tiles += ((n + 63) // 64) * ((k + 63) // 64)
            self._tr_table = torch.tensor(rows, dtype=torch.int32, device=dev) if rows else None
            self._tr_tiles = tiles
            self.sync_bf16()

    def transpose_table_for(self, start: int, end: int):
        """(device int32 table, tile count) of the 2-D tensors whose storage lies in flat elements [start, end)."""
        rows, tiles = [], 0
        for (name, p), o in zip(self.named, self.offsets):
            if p.ndim == 2 and start <= o < end:
                n, k = p.shape
                rows.append([o, n, k, tiles])
                tiles += ((n + 63) // 64) * ((k + 63) // 64)
        if not rows or self.w_bf16 is None:
            return None, 0
        return torch.tensor(rows, dtype=torch.int32, device=self.params.device), tiles

    def refresh_transposes(self):
        if self._tr_table is not None:
            ops.transpose_multi(self.w_bf16, self.wt_bf16, self._tr_table, self._tr_tiles)

    def sync_bf16(self):
        """Rebuild both bf16 copies from the fp32 master and re-arm the per-parameter views (needed after the
        parameters were written by anything other than FlatAdamW.step, which keeps the copies current itself)."""
        if self.w_bf16 is None:
            return
        self.w_bf16.copy_(ops.cast_bf16(self.params))
        self.refresh_transposes()
        for (name, p), o in zip(self.named, self.offsets):
            if p.ndim == 2:
                n, k = p.shape
                p._mmae_bf16_views = (self.w_bf16[o:o + n * k].view(n, k), self.wt_bf16[o:o + n * k].view(k, n),
                                      p._version, self.sync_bf16)

    def zero_grad(self, fill: bool = True):
        """fill=False: the buffer is known to be zero already (the bucket-wise update cleared it); only the `.grad` views are
        re-attached."""
        if fill:
            Fn.join_side_stream(self.grads.device)
            self.grads.zero_()
        for (_, p), o in zip(self.named, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * o:
                p.grad = self.grads[o:o + p.numel()].view(p.shape)  # re-attach after zero_grad(set_to_none=True)


class GradBuckets:
    """Bucketed SUM all-reduce of FlatParams.grads, launched from post-accumulate-grad hooks as buckets complete.

    Construction broadcasts rank 0's parameters (flat storage, frozen parameters, buffers) to the group, the way the
    reference's DistributedDataParallel wrap does (run_pretraining_multimae.py:224-226; the script seeds each rank with
    seed + rank before it builds the model, :147-148, so the ranks' own initialisations differ).

    `no_sync` = True suspends the exchange for gradient accumulation: hooks return at once and gradients keep
    accumulating locally; the first backward after it is cleared all-reduces the accumulated sums."""

    def __init__(self, flat: FlatParams, bucket_mb: float = 64.0, group=None, broadcast: bool = True, comm=None):
        """comm: None = collectives through torch.distributed (backend "nccl" is RCCL), or a multimae_amd.comm.NativeComm
        (the C ABI's RCCL driver: all-reduces enqueued on the communicator's own stream, ordered against the step's
        streams by events, no host-side work objects).  MMAE_COMM=native in the environment builds one."""
        self.flat, self.group = flat, group
        self.fallback_reason = None  # why the C ABI's communicator was not used although asked for
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.no_sync = False
        single = os.environ.get("MMAE_DP_SINGLE_RANK") == "1"
        if comm is None and os.environ.get("MMAE_COMM") == "native" and flat.params.is_cuda and (self.world > 1 or single):
            comm, self.group = self._native_or_fallback(group)
        self.comm = comm
        if comm is not None:
            self.world = comm.world
        if broadcast and self.world > 1:
            self.broadcast_parameters()
        cap = int(bucket_mb * (1 << 20) / 4)
        # The all-reduce of the bucket that completes last cannot hide under backward, so the buckets taper towards the
        # end of the (large) decayed region: the last one holds about cap / 4, the one before it cap / 2.
        ends = [o + (p.numel() + 7) // 8 * 8 for (_, p), o in zip(flat.named, flat.offsets)]
        taper = {}  # first element offset -> capacity, for the two tail buckets
        left = flat.n_decay
        for frac in (4, 2):
            if left > 2 * cap:
                taper[left - cap // frac] = cap // frac
                left -= cap // frac
        self.buckets: List[List[int]] = []   # [start, end, n_params]
        self.bucket_of = {}
        cur, cur_cap = None, cap
        for i, ((_, p), o) in enumerate(zip(flat.named, flat.offsets)):
            end = ends[i]
            boundary = (o == flat.n_decay)  # never mix the two regions
            start_tail = next((t for t in taper if o <= t < end), None) if o < flat.n_decay else None
            if cur is None or boundary or end - cur[0] > cur_cap or (start_tail is not None and cur[0] < start_tail - (end - o)):
                cur = [o, end, 0]
                cur_cap = taper.pop(start_tail) if start_tail is not None else cap
                self.buckets.append(cur)
            cur[1] = end
            cur[2] += 1
            self.bucket_of[i] = len(self.buckets) - 1
        self.pending = [b[2] for b in self.buckets]
        self.signalled = [False] * len(flat.named)
        # a one-rank group still runs the collectives when asked to (MMAE_DP_SINGLE_RANK=1: exercises the RCCL path on one GPU)
        self.enabled = self.world > 1 or ((dist.is_initialized() or self.comm is not None) and single)
        # called as after_bucket(b, start, end, stream) once bucket b's gradients are final (all-reduced when there is an
        # exchange): `stream` is the stream that work on the bucket must be queued on to be ordered after them
        self.after_bucket = None
        self._hooked = False
        self._comm_pending = False
        # torch.distributed path: (bucket, work) of all-reduces whose completion nothing has been ordered after yet
        self._unjoined = []
        # Measurement: `timing` = None, or a list that receives (bucket, start event, end event) around every collective
        # (the collective is then waited for on its issuing stream, so the pair brackets it; untimed steps only);
        # `exposed` receives one (start, end) event pair per step around the wait for the last collectives in finish().
        self.timing = None
        self.exposed = []
        if self.enabled:
            self.install_hooks()

    def _native_or_fallback(self, group):
        """The C ABI's RCCL communicator, or, when its set-up fails on any rank, torch.distributed's own RCCL group created
        in this process (never a re-exec: the GPU is initialised).  Returns (comm or None, group)."""
        from . import _lib
        from .comm import NativeComm
        comm, err = None, ""
        try:
            comm = NativeComm(group=group)
        except _lib.MmaeError as exc:
            err = str(exc)
        ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32)
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            if dist.get_backend(group) == "nccl":
                ok = ok.to(self.flat.params.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)  # the ranks agree on one path
        if int(ok.item()) == 1:
            return comm, group
        self.fallback_reason = err or "mmae_comm_init failed on another rank"
        print(f"[multimae_amd] native RCCL communicator unavailable ({self.fallback_reason}); using torch.distributed's",
              flush=True)
        if comm is not None:
            comm.destroy()
        if not dist.is_initialized():
            raise _lib.MmaeError("native communicator failed and there is no torch.distributed group to fall back to: " + err)
        if dist.get_backend(group) != "nccl":  # the host channel is gloo: a RCCL group over the same ranks
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            group = dist.new_group(ranks=ranks, backend="nccl", device_id=self.flat.params.device)
        return None, group

    def describe(self):
        """What the exchange is made of, for logs and bench lines."""
        if self.comm is not None:
            backend, seen = "native", self.comm.info()[1]
        elif dist.is_initialized():
            backend, seen = "torch-" + dist.get_backend(self.group), dist.get_world_size(self.group)
        else:
            backend, seen = "none", 1
        return {"backend": backend, "ranks_seen": seen, "world": self.world, "enabled": bool(self.enabled),
                "buckets": len(self.buckets), "bucket_bytes": [4 * (e - s) for s, e, _ in self.buckets],
                "bytes_per_step": 4 * sum(e - s for s, e, _ in self.buckets), "fallback_reason": self.fallback_reason}

    def install_hooks(self):
        """Per-parameter completion signals (idempotent): post-accumulate hooks for gradients delivered by autograd,
        `_mmae_ready` callbacks for gradients the kernels wrote in place."""
        if self._hooked:
            return
        self._hooked = True
        for i, (_, p) in enumerate(self.flat.named):
            hook = self._make_hook(i)
            p.register_post_accumulate_grad_hook(hook)      # gradients delivered by autograd
            p._mmae_ready = (lambda h=hook, q=p: h(q))       # gradients the kernels wrote in place

    def broadcast_parameters(self, src: int = 0):
        """Every rank takes rank `src`'s parameters and buffers; the bf16 weight copies are rebuilt from them."""
        with torch.no_grad():
            if self.comm is not None:
                cur = torch.cuda.current_stream(self.flat.params.device)
                self.comm.stream.wait_stream(cur)
                for t in [self.flat.params] + [f.data for f in self.flat.frozen]:
                    if t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous():
                        self.comm.broadcast_(t, root=src)
                    else:  # other dtypes / layouts: through a float32 copy
                        tmp = t.to(torch.float32).contiguous()
                        self.comm.broadcast_(tmp, root=src)
                        tmp.record_stream(self.comm.stream)
                        with torch.cuda.stream(self.comm.stream):
                            t.copy_(tmp)
                cur.wait_stream(self.comm.stream)
            else:
                src_global = dist.get_global_rank(self.group, src) if self.group is not None else src
                staged = self.flat.params.is_cuda and dist.get_backend(self.group) == "gloo"  # see _launch
                for t in [self.flat.params] + [f.data for f in self.flat.frozen]:
                    if staged:
                        host = t.cpu()
                        dist.broadcast(host, src=src_global, group=self.group)
                        t.copy_(host)
                    else:
                        dist.broadcast(t, src=src_global, group=self.group)
        self.flat.sync_bf16()

    def _make_hook(self, i):
        def hook(_p):
            if self.no_sync:
                return
            # A parameter signals once per step.  Autograd also runs the post-accumulate hook of a parameter whose
            # gradient a kernel wrote in place (the Function returned None for it), after the kernel's own signal.
            if self.signalled[i]:
                return
            self.signalled[i] = True
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        """Bucket b is complete: every kernel that writes it has been enqueued, on the current stream, a decoder stream or
        the weight-gradient stream (functional._lin_grads).  All-reduce it (when there is an exchange) from a stream that
        waits for all of them, then hand it to `after_bucket`."""
        s, e, _ = self.buckets[b]
        buf = self.flat.grads[s:e]
        if buf.is_cuda:  # kernels of several decoder streams may have written into this bucket
            Fn.join_aux_streams(buf.device, reset=False)
        side = Fn.side_stream_pending(buf.device) if buf.is_cuda else None
        if self.comm is not None and self.enabled:
            # ordered after everything queued so far on the streams that write gradients, and enqueued on the communicator's
            # own stream: the host neither waits nor keeps a handle per collective
            self.comm.stream.wait_stream(torch.cuda.current_stream(buf.device))
            if side is not None:
                self.comm.stream.wait_stream(side)
            if self.timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.comm.stream)
            self.comm.all_reduce_(buf)
            if self.timing is not None:
                e1.record(self.comm.stream)
                self.timing.append((b, e0, e1))
            self._comm_pending = True
            if self.after_bucket is not None:
                self.after_bucket(b, s, e, self.comm.stream)
            return
        if not self.enabled:  # one rank, nothing to exchange: the bucket only goes to after_bucket
            if self.after_bucket is not None:
                if buf.is_cuda and Fn.SIDE_STREAM_DW:
                    ent = Fn._side(buf.device)
                    ent[0].wait_stream(torch.cuda.current_stream(buf.device))
                    self.after_bucket(b, s, e, ent[0])
                    ent[1] = True
                else:
                    self.after_bucket(b, s, e, torch.cuda.current_stream(buf.device) if buf.is_cuda else None)
            return
        # gloo is a rehearsal backend here (several ranks on one GPU, which RCCL refuses).  Given a device tensor it stages it
        # through the host on high-priority streams of its own that wait on this step's streams by events, and a
        # high-priority stream waiting for pending work of ordinary streams can stop the process's queues for good on this
        # ROCm stack (scripts/prio_inversion.py); draining the device before every collective made that rare, not
        # impossible (two ranks of the ViT-B bench on one GPU still stalled in the wait for a collective in two runs of
        # five).  So gloo only ever sees host tensors: the bucket is copied to the host on the current stream, reduced
        # there and copied back, one collective at a time.  RCCL (ordinary-priority streams) stays asynchronous.
        if buf.is_cuda and dist.get_backend(self.group) == "gloo":
            if side is not None:
                torch.cuda.current_stream(buf.device).wait_stream(side)
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            buf.copy_(host)
            return
        # RCCL through torch.distributed: the collective runs on the process group's own stream, ordered after the stream
        # it is issued from.  `work.wait()` makes the CURRENT stream wait for it (the host does not wait).
        if not buf.is_cuda:
            self._unjoined.append((b, dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
            return
        issue = side if side is not None else torch.cuda.current_stream(buf.device)
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(buf.device))
        with torch.cuda.stream(issue):
            if self.timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(issue)
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if self.timing is not None:
                work.wait()
                e1.record(issue)
                self.timing.append((b, e0, e1))
        self._unjoined.append((b, work))
        # With a bucket-wise consumer (the optimizer) a bucket is handed over one bucket late: by the time the next bucket
        # is complete its all-reduce has had a bucket's worth of backward to finish under, so the stream that takes the
        # update (the weight-gradient stream when there is one) hardly ever stalls in the wait.
        if self.after_bucket is not None:
            while len(self._unjoined) > 1:
                self._hand_over(*self._unjoined.pop(0))

    def _hand_over(self, b, work, final=False):
        """Order the consumer of bucket b after its all-reduce and run it: on the weight-gradient stream while backward is
        still going, on the current stream at the end of the step."""
        s, e, _ = self.buckets[b]
        dev = self.flat.grads.device
        cur = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        if final or cur is None or not Fn.SIDE_STREAM_DW:
            work.wait()
            if self.after_bucket is not None:
                self.after_bucket(b, s, e, cur)
            return
        ent = Fn._side(dev)
        ent[0].wait_stream(cur)  # the update writes the bf16 weight copies: after every product of backward queued so far
        with torch.cuda.stream(ent[0]):
            work.wait()
        self.after_bucket(b, s, e, ent[0])
        ent[1] = True

    def reset(self):
        """Forget the completion signals collected so far (nothing may be in flight)."""
        if not self._unjoined and not self.no_sync:
            self.pending = [b[2] for b in self.buckets]
            self.signalled = [False] * len(self.signalled)

    def finish(self):
        """Call after backward: launches any bucket whose parameters received no gradient, waits for all."""
        if not self._hooked or self.no_sync:
            return
        for b, left in enumerate(self.pending):
            if left > 0:
                self._launch(b)
        waiting = bool(self._unjoined) or (self.comm is not None and self._comm_pending)
        if waiting and self.flat.grads.is_cuda:
            # how long the step's stream sits in the wait for the last collectives: the part of the exchange that backward
            # did not hide (the pair is read after the run; nothing here waits on the host)
            cur = torch.cuda.current_stream(self.flat.grads.device)
            Fn.join_side_stream(self.flat.grads.device)
            x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            x0.record(cur)
        for b, work in self._unjoined:
            self._hand_over(b, work, final=True)
        self._unjoined = []
        if self.comm is not None and self._comm_pending:
            torch.cuda.current_stream(self.flat.grads.device).wait_stream(self.comm.stream)
            self._comm_pending = False
        if waiting and self.flat.grads.is_cuda:
            x1.record(cur)
            self.exposed.append((x0, x1))
            del self.exposed[:-256]
        self.pending = [b[2] for b in self.buckets]
        self.signalled = [False] * len(self.signalled)


class FlatAdamW:
    """AdamW over FlatParams in one kernel launch (PyTorch AdamW semantics; betas / weight decay as the reference's
    pretraining defaults, pretrain_argparser.py:44-64 there), with the gradient clipping / skipping of the reference's
    NativeScalerWithGradNormCount (utils/native_scaler.py:33-51) decided on the device."""

    def __init__(self, flat: FlatParams, lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05):
        self.flat = flat
        # the two groups of the flat layout, writable like torch's (the reference's loop assigns "lr" and
        # "weight_decay" per iteration, run_pretraining_multimae.py:427-436); one learning rate serves both
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "lr_scale": 1.0, "betas": betas, "eps": eps},
                             {"lr": lr, "weight_decay": 0.0, "lr_scale": 1.0, "betas": betas, "eps": eps}]
        self.betas, self.eps = betas, eps
        self.exp_avg = torch.zeros_like(flat.params)
        self.exp_avg_sq = torch.zeros_like(flat.params)
        self.step_count = 0   # calls of step()
        self.step_dev = None  # applied updates (device int32), created on the first step that may skip

    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, v: float):
        for g in self.param_groups:
            g["lr"] = v

    @property
    def weight_decay(self) -> float:
        return self.param_groups[0]["weight_decay"]

    @weight_decay.setter
    def weight_decay(self, v: float):
        self.param_groups[0]["weight_decay"] = v

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()

    def step(self, grad_scale: float = 1.0, clip_grad: Optional[float] = None, skip_grad: Optional[float] = None,
             grad_sumsq: Optional[torch.Tensor] = None):
        """grad_sumsq: device scalar sum(g^2) of the raw gradient buffer (computed here if clipping / skipping needs it).
        With skip_grad the number of applied updates lives on the device (no host synchronisation).
        As in the reference's scaler (utils/native_scaler.py:35-46) clipping takes precedence: with clip_grad given
        (None = off; any number, 0.0 included, clips) skip_grad is ignored."""
        if self.param_groups[1]["lr"] != self.param_groups[0]["lr"]:
            raise ValueError("FlatAdamW applies one learning rate to both parameter groups")
        Fn.join_side_stream(self.flat.grads.device)
        self.step_count += 1
        if clip_grad is not None:
            skip_grad = None
            clip_grad = max(float(clip_grad), 1e-30)  # the kernel reads 0 as "no clipping"; a zero bound scales to nothing
        elif skip_grad is not None and not skip_grad > 0:
            skip_grad = 1e-30                         # norm >= 0 always holds: every step is skipped, as in the reference
        if (clip_grad is not None or skip_grad is not None) and grad_sumsq is None:
            grad_sumsq = self.grad_norm_sq()
        if skip_grad is not None and self.step_dev is None:
            self.step_dev = torch.full((1,), self.step_count - 1, dtype=torch.int32, device=self.flat.params.device)
        ops.adamw_flat(self.flat.params, self.flat.grads, self.exp_avg, self.exp_avg_sq, self.flat.n_decay, self.lr,
                       self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, grad_scale,
                       w_bf16=self.flat.w_bf16, grad_sumsq=grad_sumsq if (clip_grad is not None or skip_grad is not None) else None,
                       max_norm=clip_grad or 0.0, skip_norm=skip_grad or 0.0, step_dev=self.step_dev)
        self.flat.refresh_transposes()
        Fn.bump_weight_epoch()

    def begin_step(self):
        """Start an optimisation step whose update is applied range by range (step_range) while backward still runs."""
        if self.param_groups[1]["lr"] != self.param_groups[0]["lr"]:
            raise ValueError("FlatAdamW applies one learning rate to both parameter groups")
        self.step_count += 1

    def step_range(self, start: int, end: int, grad_scale: float = 1.0, transpose_table=None, clear_grad=False, sumsq_part=None):
        """The update of begin_step()'s step for flat elements [start, end) (a range that does not straddle the decayed /
        not decayed boundary), on the current stream; `transpose_table` = (int32 table, tiles) of the 2-D weights inside
        the range refreshes their transposed bf16 copies.  No clipping / skipping: those need the whole gradient first.
        clear_grad: the range's gradients are zeroed behind the read (the next iteration's zero_grad); sumsq_part: receives
        ops.adamw_parts(end - start) partial sums of the range's squared gradients (see ops.adamw_flat)."""
        nd = self.flat.n_decay
        if start < nd < end:
            raise ValueError("step_range: the range straddles the weight-decay boundary")
        f = self.flat
        ops.adamw_flat(f.params[start:end], f.grads[start:end], self.exp_avg[start:end], self.exp_avg_sq[start:end],
                       end - start if start < nd else 0, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                       self.step_count, grad_scale, w_bf16=None if f.w_bf16 is None else f.w_bf16[start:end],
                       clear_grad=clear_grad, sumsq_part=sumsq_part)
        if transpose_table is not None and transpose_table[0] is not None:
            ops.transpose_multi(f.w_bf16, f.wt_bf16, transpose_table[0], transpose_table[1])

    def end_step(self):
        Fn.bump_weight_epoch()

    def grad_norm_sq(self) -> torch.Tensor:
        Fn.join_side_stream(self.flat.grads.device)
        return ops.sumsq(self.flat.grads)

    def state_dict(self):
        applied = int(self.step_dev.item()) if self.step_dev is not None else self.step_count
        return {"step": applied, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr,
                "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.step_dev = None
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


def cosine_scheduler(base_value: float, final_value: float, epochs: int, niter_per_ep: int, warmup_epochs=0,
                     start_warmup_value=0, warmup_steps=-1):
    """Per-iteration values: linear warm-up then half-cosine decay (reference: utils/native_scaler.py:86-117, same
    arithmetic: the warm-up ramp exists only when warmup_epochs > 0, its length is warmup_steps when that is > 0)."""
    import math
    import numpy as np
    warmup_iters = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        warmup_iters = warmup_steps
    warmup = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    n = epochs * niter_per_ep - warmup_iters
    sched = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / n)) for i in range(n)])
    sched = np.concatenate((warmup, sched))
    assert len(sched) == epochs * niter_per_ep
    return sched


class PretrainStep:
    """model + losses + flat AdamW + gradient buckets: one call = one optimisation step on a batch already on the device."""

    def __init__(self, model, losses, num_encoded_tokens=98, alphas=1.0, lr=1e-4, weight_decay=0.05, betas=(0.9, 0.95),
                 bucket_mb=64.0, clip_grad=None, skip_grad=None, lr_schedule_values=None, wd_schedule_values=None,
                 start_step=0, sample_tasks_uniformly=False, fp32_output_adapters=(), standardize_depth=False,
                 eager_optimizer=None, max_steps_in_flight=None, loss_on_unmasked=False, keep_grads=False):
        """clip_grad / skip_grad: max_norm / max_skip_norm of the reference's train_one_epoch
        (run_pretraining_multimae.py:512-519).  lr_schedule_values / wd_schedule_values: per-iteration arrays
        (cosine_scheduler) applied before each step as at :424-436; iteration = start_step + calls so far.
        loss_on_unmasked: the losses average over every token instead of the masked ones (:491-496).
        keep_grads: leave the step's gradients in `p.grad` after the call (for inspection).  By default the bucket-wise
        update clears each gradient range behind its read, which is the next iteration's `optimizer.zero_grad()` (:507)
        done early; with clipping / skipping (update at the end of the step) the gradients stay either way."""
        self.model, self.losses = model, losses
        self.k, self.alphas = num_encoded_tokens, alphas
        self.clip_grad, self.skip_grad = clip_grad, skip_grad
        self.lr_values, self.wd_values, self.it = lr_schedule_values, wd_schedule_values, start_step
        self.uniform, self.fp32_out = sample_tasks_uniformly, list(fp32_output_adapters)
        self.loss_on_unmasked = bool(loss_on_unmasked)
        self.standardize_depth = standardize_depth  # truncated standardisation of batch["depth"] (reference :448-459)
        self.flat = FlatParams(model)
        self.buckets = GradBuckets(self.flat, bucket_mb=float(os.environ.get("MMAE_BUCKET_MB", bucket_mb)))  # env: tuning
        self.opt = FlatAdamW(self.flat, lr=lr, betas=betas, weight_decay=weight_decay)
        self.world = self.buckets.world
        # The optimizer is bandwidth-bound work at the end of a step of matrix products.  Without clipping / skipping (which
        # need the norm of the whole gradient first; the reference's default, pretrain_argparser.py:47-50) a bucket of
        # parameters can be updated as soon as its gradients are final, beside the rest of backward: on the weight-gradient
        # stream with one rank, on the communicator's stream behind the bucket's all-reduce with the C ABI's RCCL driver, and
        # with torch.distributed's RCCL group on the weight-gradient stream one bucket late (GradBuckets._hand_over: the
        # group's collectives run on a stream of its own that nothing else can be queued on, so the update waits for it from
        # another stream, at a moment when the all-reduce has normally finished).  gloo (host-staged, rehearsal) keeps the
        # update at the end of the step.
        if eager_optimizer is None:
            eager_optimizer = os.environ.get("MMAE_EAGER_OPT", "1") != "0"
        staged = (self.buckets.enabled and self.buckets.comm is None and dist.is_initialized()
                  and dist.get_backend(self.buckets.group) != "nccl")
        self.eager = bool(eager_optimizer and clip_grad is None and skip_grad is None and self.flat.params.is_cuda
                          and not staged)
        if self.eager:
            self._tr_tables = [self.flat.transpose_table_for(s, e) for s, e, _ in self.buckets.buckets]
            self.buckets.install_hooks()
            self.buckets.after_bucket = self._update_bucket
            # The bucket-wise update reads every gradient exactly once, after it is final: the same read leaves the partial
            # sums of the squared gradient norm and clears the gradients for the next step, so neither a norm pass nor a
            # fill pass over the 0.4-GB buffer runs between two steps.
            counts = [ops.adamw_parts(e - s) for s, e, _ in self.buckets.buckets]
            self._part_ranges = [(sum(counts[:b]), sum(counts[:b + 1])) for b in range(len(counts))]
            self._norm_parts = torch.zeros(sum(counts), dtype=torch.float32, device=self.flat.params.device)
        self.keep_grads = bool(keep_grads)
        self._fused_norm = os.environ.get("MMAE_FUSED_NORM", "1") != "0"  # tuning: 0 = separate norm and fill passes
        self._grads_clean = False  # True after a step in which every bucket was updated (and thereby cleared)
        self._updates = 0
        self._armed = False
        # The host enqueues a step several times faster than the device runs it and never has to wait for it, so without a
        # bound it runs many steps ahead.  Memory then grows with the distance: buffers that were used on a second stream are
        # only reusable once that stream has passed them (record_stream), and every step in flight holds its own set (ViT-L/16
        # 384^2 at 256 images: step times of 0.3-0.8 s instead of 0.2 s from allocator traffic, then a queue abort with
        # HSA_STATUS_ERROR_OUT_OF_RESOURCES).  Before enqueueing step i the host therefore waits for step
        # i - max_steps_in_flight to have finished: the queues stay at least one whole step deep, the device never idles.
        if max_steps_in_flight is None:
            max_steps_in_flight = int(os.environ.get("MMAE_MAX_STEPS_IN_FLIGHT", "2"))
        self.max_in_flight = max_steps_in_flight if self.flat.params.is_cuda else 0
        self._step_events = []

    def _update_bucket(self, b, start, end, stream):
        if not self._armed:  # a backward pass through the model outside this step (tests, probes): gradients only
            self._grads_clean = False
            return
        lo, hi = self._part_ranges[b]
        with torch.cuda.stream(stream):
            self.opt.step_range(start, end, grad_scale=1.0 / self.world, transpose_table=self._tr_tables[b],
                                clear_grad=self._fused_norm and not self.keep_grads,
                                sumsq_part=self._norm_parts[lo:hi] if self._fused_norm else None)
        self._updates += 1

    def __call__(self, batch, mask_randomness=None):
        if self.lr_values is not None:
            self.opt.lr = float(self.lr_values[self.it])
        if self.wd_values is not None and self.opt.weight_decay > 0:
            self.opt.weight_decay = float(self.wd_values[self.it])
        self.it += 1
        if self.max_in_flight > 0:
            while len(self._step_events) >= self.max_in_flight:
                ev = self._step_events.pop(0)
                if not ev.query():
                    ev.synchronize()
        self.flat.zero_grad(fill=not (self.eager and self._grads_clean))
        self._grads_clean, self._updates = False, 0
        if self.eager:
            self.opt.begin_step()
        if self.standardize_depth and "depth" in batch:
            batch = dict(batch)
            batch["depth"] = ops.standardize_depth(batch["depth"].contiguous())
        inputs = {d: t for d, t in batch.items() if d in self.model.input_adapters}
        preds, masks = self.model(inputs, num_encoded_tokens=self.k, alphas=self.alphas, mask_randomness=mask_randomness,
                                  sample_tasks_uniformly=self.uniform, fp32_output_adapters=self.fp32_out)
        task_losses = {t: fn(preds[t], batch[tgt], mask=None if self.loss_on_unmasked else masks.get(tgt))
                       for t, (fn, tgt) in self.losses.items()}
        loss = sum(task_losses.values())
        Fn.mark_unit_upstream(task_losses.values())  # loss is their plain sum: d(loss) / d(task loss) = 1 for every task
        self.buckets.reset()        # completion signals left by a backward pass outside this step
        self._armed = True
        try:
            loss.backward()
            self.buckets.finish()
        finally:
            self._armed = False
        Fn.join_side_stream()
        # squared norm of the summed (not yet averaged) gradients: norm = sqrt(gn2) / world
        if self.eager and self._fused_norm and self._updates == len(self.buckets.buckets):
            gn2 = ops.sum_partials(self._norm_parts)  # left by the bucket updates, which also cleared the gradients
            self._grads_clean = not self.keep_grads
        else:
            gn2 = self.opt.grad_norm_sq()
        if self.eager:
            self.opt.end_step()  # every bucket was updated as it completed
        else:
            self.opt.step(grad_scale=1.0 / self.world, clip_grad=self.clip_grad, skip_grad=self.skip_grad, grad_sumsq=gn2)
        if self.max_in_flight > 0:
            ev = torch.cuda.Event()
            ev.record()
            self._step_events.append(ev)
        return loss.detach(), task_losses, gn2
