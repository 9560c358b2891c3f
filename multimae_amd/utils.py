"""The helpers the reference's training script takes from its ``utils`` package for the step itself
(run_pretraining_multimae.py:38,45-46,53,144,180-181,228,233,262,330,411-412 there), over this package's runtime.
Data loading and argument parsing are not part of the step and are not provided; the meters the loop wraps around
the step (MetricLogger, SmoothedValue, a WandbLogger without a back end) are in .logger."""
from typing import Iterator, Optional

import torch
import torch.distributed as dist

from . import checkpoint as _ckpt
from .checkpoint import save_model_for_downstream_tasks  # noqa: F401
from .logger import MetricLogger, SmoothedValue, WandbLogger  # noqa: F401
from .multimae import create_model  # noqa: F401
from .train import FlatAdamW, FlatParams, GradBuckets, cosine_scheduler, init_distributed_mode, init_from_env  # noqa: F401


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process() -> bool:
    return get_rank() == 0


def create_optimizer(args_or_model, model=None, lr: Optional[float] = None, weight_decay: Optional[float] = None, betas=None,
                     eps: Optional[float] = None, bucket_mb: float = 64.0):
    """AdamW over flat storage + the data-parallel gradient buckets for a model (the role of
    utils.optim_factory.create_optimizer + the DistributedDataParallel wrap at run_pretraining_multimae.py:224-228).
    Call after the model is on its GPU.  Two call shapes:

      create_optimizer(args, {"model": model})   the reference's (:228): lr, weight_decay, opt_betas and opt_eps are read
          from `args` (pretrain_argparser.py:44-64 there; lr falls back to blr, the script sets args.lr = blr * batch / 256
          itself at :213-214); returns the optimizer, which carries its GradBuckets as `.buckets` (a
          NativeScalerWithGradNormCount built without buckets picks them up from there);
      create_optimizer(model, lr=..., weight_decay=...)   returns (optimizer, buckets)."""
    if isinstance(args_or_model, torch.nn.Module):
        net, ref_shape = args_or_model, False
        lr = 1e-4 if lr is None else lr
        weight_decay = 0.05 if weight_decay is None else weight_decay
        betas = (0.9, 0.95) if betas is None else tuple(betas)
        eps = 1e-8 if eps is None else eps
    else:
        args, ref_shape = args_or_model, True
        net = model["model"] if isinstance(model, dict) else model
        if not isinstance(net, torch.nn.Module):
            raise TypeError('create_optimizer(args, {"model": model}): no model given')
        opt_name = str(getattr(args, "opt", "adamw") or "adamw").lower()
        if opt_name != "adamw":
            raise NotImplementedError(f"optimizer '{opt_name}': only AdamW (the reference's pretraining default) is implemented")
        if lr is None:
            lr = getattr(args, "lr", None)
            lr = getattr(args, "blr", 1e-4) if lr is None else lr
        weight_decay = getattr(args, "weight_decay", 0.05) if weight_decay is None else weight_decay
        if betas is None:
            betas = getattr(args, "opt_betas", None) or (0.9, 0.95)
        if eps is None:
            eps = getattr(args, "opt_eps", None) or 1e-8
    flat = FlatParams(net)
    opt = FlatAdamW(flat, lr=float(lr), betas=tuple(float(b) for b in betas), eps=float(eps), weight_decay=float(weight_decay))
    opt.buckets = GradBuckets(flat, bucket_mb=bucket_mb)
    return opt if ref_shape else (opt, opt.buckets)


class NativeScalerWithGradNormCount:
    """Call-compatible with the reference's loss scaler (utils/native_scaler.py:17-61): backward, gradient norm,
    optional clipping / skipping (clipping takes precedence when both are given, :35-46), optimizer step.  bf16 needs no
    loss scale, so the "scale" it reports is 1.0.  `optimizer` is a FlatAdamW; its GradBuckets come from the
    constructor argument or from `optimizer.buckets` (set by create_optimizer).  update_grad=False accumulates: backward
    only, with the gradient exchange suspended until the call that updates."""
    state_dict_key = "amp_scaler"

    def __init__(self, enabled: bool = True, buckets: Optional[GradBuckets] = None):
        self.buckets = buckets

    def __call__(self, loss: torch.Tensor, optimizer: FlatAdamW, clip_grad: Optional[float] = None,
                 skip_grad: Optional[float] = None, parameters: Optional[Iterator] = None, create_graph: bool = False,
                 update_grad: bool = True):
        buckets = self.buckets if self.buckets is not None else getattr(optimizer, "buckets", None)
        if buckets is not None:
            buckets.no_sync = not update_grad
        loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        world = 1
        if buckets is not None:
            buckets.finish()
            world = buckets.world
        gn2 = optimizer.grad_norm_sq()
        optimizer.step(grad_scale=1.0 / world, clip_grad=clip_grad, skip_grad=skip_grad, grad_sumsq=gn2)
        return gn2.sqrt() / world

    def state_dict(self):
        return {"scale": 1.0}

    def load_state_dict(self, state_dict):
        pass


def plain_args(args):
    """The run configuration as a dict of plain values: what checkpoint["args"] holds here.  The reference pickles the
    argument object itself (utils/checkpoint.py:122 there), which a weights_only load refuses; the fields a resumed run
    reads (numbers, strings, flags, lists of them) survive as they are."""
    plain = (int, float, str, bool, type(None))
    out = {}
    for k, v in vars(args).items():
        if k.startswith("_"):
            continue
        if isinstance(v, plain):
            out[k] = v
        elif isinstance(v, (list, tuple)) and all(isinstance(e, plain) for e in v):
            out[k] = list(v)
    return out


def save_model(*pos, args=None, epoch=None, model=None, model_without_ddp=None, optimizer=None, loss_scaler=None,
               model_ema=None, **kw):
    """checkpoint.save_model, also in the reference's call shape save_model(args=, epoch=, model=, model_without_ddp=,
    optimizer=, loss_scaler=) (utils/checkpoint.py:103-131 there): args.output_dir names the directory, and the
    encoder-side export for downstream tasks is written next to the checkpoint (:129-130)."""
    if pos or args is None or not hasattr(args, "output_dir"):
        return _ckpt.save_model(*pos, **{k: v for k, v in dict(epoch=epoch, model=model, optimizer=optimizer, args=args).items()
                                         if v is not None}, **kw)
    if model_ema is not None:
        raise NotImplementedError("model_ema is not part of the pretraining step")
    net = model_without_ddp if model_without_ddp is not None else model
    scaler_state = loss_scaler.state_dict() if loss_scaler is not None else None
    path = _ckpt.save_model(str(args.output_dir), epoch, net, optimizer, args=plain_args(args), scaler_state=scaler_state)
    _ckpt.save_model_for_downstream_tasks(str(args.output_dir), str(getattr(args, "wandb_run_name", "run")), epoch, net)
    return path


def auto_load_model(*pos, args=None, model=None, model_without_ddp=None, optimizer=None, loss_scaler=None, model_ema=None, **kw):
    """checkpoint.auto_load_model, also in the reference's call shape auto_load_model(args=, model=, model_without_ddp=,
    optimizer=, loss_scaler=) (utils/checkpoint.py:162-206 there): reads args.output_dir / resume / auto_resume /
    pretrained_weights and sets args.resume and args.start_epoch as the reference does."""
    if pos or args is None or not hasattr(args, "output_dir"):
        return _ckpt.auto_load_model(*pos, **{k: v for k, v in dict(model=model, optimizer=optimizer).items() if v is not None}, **kw)
    net = model_without_ddp if model_without_ddp is not None else model
    resume = getattr(args, "resume", "") or ""
    if str(resume).startswith("https"):
        raise NotImplementedError("resuming from a URL needs a network")
    if getattr(args, "auto_resume", True) and not resume:
        resume = _ckpt.latest_checkpoint(str(args.output_dir)) or ""
        args.resume = resume
    start = _ckpt.auto_load_model(str(args.output_dir), net, optimizer, resume=resume, auto_resume=False,
                                  pretrained_weights=getattr(args, "pretrained_weights", None), **kw)
    if start > 0:
        args.start_epoch = start
    return start
