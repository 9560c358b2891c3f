"""The helpers the reference's training script takes from its ``utils`` package for the step itself
(run_pretraining_multimae.py:38,45-46,144,180-181,233,262,330,411-412 there), over this package's runtime.
Logging, data loading and argument parsing are not part of the step and are not provided."""
from typing import Iterator, Optional

import torch
import torch.distributed as dist

from .checkpoint import auto_load_model, save_model, save_model_for_downstream_tasks  # noqa: F401
from .multimae import create_model  # noqa: F401
from .train import FlatAdamW, FlatParams, GradBuckets, cosine_scheduler, init_distributed_mode  # noqa: F401


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process() -> bool:
    return get_rank() == 0


def create_optimizer(model: torch.nn.Module, lr: float = 1e-4, weight_decay: float = 0.05, betas=(0.9, 0.95), eps: float = 1e-8,
                     bucket_mb: float = 64.0):
    """AdamW over flat storage + the data-parallel gradient buckets for `model` (the role of
    utils.optim_factory.create_optimizer + the DistributedDataParallel wrap at run_pretraining_multimae.py:224-228).
    Returns (optimizer, buckets); call after the model is on its GPU."""
    flat = FlatParams(model)
    return FlatAdamW(flat, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay), GradBuckets(flat, bucket_mb=bucket_mb)


class NativeScalerWithGradNormCount:
    """Call-compatible with the reference's loss scaler (utils/native_scaler.py:17-61): backward, gradient norm,
    optional clipping / skipping, optimizer step.  bf16 needs no loss scale, so the "scale" it reports is 1.0.
    `optimizer` is a FlatAdamW; pass the GradBuckets at construction when training data-parallel."""
    state_dict_key = "amp_scaler"

    def __init__(self, enabled: bool = True, buckets: Optional[GradBuckets] = None):
        self.buckets = buckets

    def __call__(self, loss: torch.Tensor, optimizer: FlatAdamW, clip_grad: Optional[float] = None,
                 skip_grad: Optional[float] = None, parameters: Optional[Iterator] = None, create_graph: bool = False,
                 update_grad: bool = True):
        loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        world = 1
        if self.buckets is not None:
            self.buckets.finish()
            world = self.buckets.world
        gn2 = optimizer.grad_norm_sq()
        optimizer.step(grad_scale=1.0 / world, clip_grad=clip_grad, skip_grad=skip_grad, grad_sumsq=gn2)
        return gn2.sqrt() / world

    def state_dict(self):
        return {"scale": 1.0}

    def load_state_dict(self, state_dict):
        pass
