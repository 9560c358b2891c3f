"""RCCL communicator driven through the C ABI (mmae_comm_* in include/mmae.h): the gradient exchange of the data-parallel
step on raw device pointers and an explicit HIP stream, without torch.distributed's work objects.

The reference gets the same exchange from DistributedDataParallel over the NCCL backend
(run_pretraining_multimae.py:224-226, utils/dist.py:94-107).  torch.distributed is used here only as the host channel that
carries rank 0's 128-byte communicator id to the other ranks."""
import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib

_DTYPES = {torch.float32: 0, torch.bfloat16: 1}


class NativeComm:
    """One RCCL communicator for this process's current GPU plus a dedicated stream for its collectives."""

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, group=None):
        lib = _lib.load()
        if rank is None or world is None:
            if dist.is_initialized():
                rank, world = dist.get_rank(group), dist.get_world_size(group)
            else:
                rank, world = 0, 1
        self.rank, self.world = int(rank), int(world)
        # One RCCL communicator per process and GPU: beside torch.distributed's own "nccl" group on the same device a
        # barrier was seen to hang (both communicators' kernels wait for their peers on the process's four hardware
        # queues).  train.init_distributed_mode starts torch.distributed over gloo when MMAE_COMM=native for this reason.
        if dist.is_initialized() and dist.get_backend(group) == "nccl":
            raise _lib.MmaeError("NativeComm: torch.distributed already runs an RCCL ('nccl') group in this process; start "
                                 "it with backend 'gloo' (the host channel for the communicator id) or use that group")
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(lib.mmae_comm_unique_id(ident), "comm_unique_id")
        if self.world > 1:
            if not dist.is_initialized():
                raise _lib.MmaeError("NativeComm: more than one rank needs an initialised torch.distributed group to hand the "
                                     "communicator id from rank 0 to the others")
            box = [ident.raw if self.rank == 0 else None]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            ident = C.create_string_buffer(box[0], 128)
        handle = C.c_void_p()
        _lib.check(lib.mmae_comm_init(ident, self.rank, self.world, C.byref(handle)), "comm_init")
        self._lib, self._handle = lib, handle
        # An ordinary-priority stream on purpose.  On this ROCm stack a high-priority stream that waits on events recorded
        # on ordinary streams of the same process can stop every queue of the process for good (scripts/prio_inversion.py
        # shows it with idle kernels alone), and the collectives here wait on the step's compute streams by events.
        self.stream = torch.cuda.Stream()

    def all_reduce_(self, t: torch.Tensor, stream: Optional[torch.cuda.Stream] = None):
        """In-place SUM over the ranks, enqueued on `stream` (default: the communicator's own stream)."""
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in _DTYPES:
            raise _lib.MmaeError("comm all_reduce: expected a contiguous f32 / bf16 tensor on the GPU")
        st = stream if stream is not None else self.stream
        _lib.check(self._lib.mmae_comm_allreduce_bucket(self._handle, C.c_void_p(t.data_ptr()), t.numel(), _DTYPES[t.dtype],
                                                        C.c_void_p(st.cuda_stream)), "comm_allreduce_bucket")

    def broadcast_(self, t: torch.Tensor, root: int = 0, stream: Optional[torch.cuda.Stream] = None):
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in _DTYPES:
            raise _lib.MmaeError("comm broadcast: expected a contiguous f32 / bf16 tensor on the GPU")
        st = stream if stream is not None else self.stream
        _lib.check(self._lib.mmae_comm_broadcast(self._handle, C.c_void_p(t.data_ptr()), t.numel(), _DTYPES[t.dtype], int(root),
                                                 C.c_void_p(st.cuda_stream)), "comm_broadcast")

    def info(self):
        """(rank, rank count) as the RCCL communicator itself reports them."""
        r, w = C.c_int(-1), C.c_int(-1)
        _lib.check(self._lib.mmae_comm_info(self._handle, C.byref(r), C.byref(w)), "comm_info")
        return r.value, w.value

    def destroy(self):
        if self._handle is not None:
            torch.cuda.synchronize()
            _lib.check(self._lib.mmae_comm_destroy(self._handle), "comm_destroy")
            self._handle = None
