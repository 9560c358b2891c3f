"""Checkpoint interchange with the reference's on-disk format (utils/checkpoint.py:103-206 there).

A training checkpoint is ``checkpoint-<epoch>.pth`` = {"model": state_dict, "optimizer": torch.optim.AdamW-style
state_dict, "epoch": int, "scaler": {...}, "args": ...}; the downstream export is
``multimae_<run>_e<epoch>.pth`` = OrderedDict(model=<encoder-side tensors>) with exactly the keys the reference
lists in utils/multimae_keys.py (global tokens, input adapters, encoder blocks: everything but the decoders).
Parameter names and shapes are the reference's, so either side loads the other's "model" entry with strict=True."""
import glob
import os
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from .train import FlatAdamW, split_decay


def _is_main_process() -> bool:
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def _save_on_master(obj, path: str):
    """Rank 0 writes (reference: save_on_master, utils/checkpoint.py:128-131 there) to a temporary file that is renamed
    into place, so a reader (latest_checkpoint on another rank, a resume after a crash) never sees a partial file;
    every rank then waits for the write.  A failed write (disk full, permissions) removes the temporary file and raises
    on EVERY rank: rank 0 still reaches the exchange below, so nobody is left waiting in it."""
    error = None
    if _is_main_process():
        tmp = path + ".tmp.%d" % os.getpid()
        try:
            torch.save(obj, tmp)
            os.replace(tmp, path)
        except Exception as exc:  # noqa: BLE001 - reported to every rank below
            error = exc
            try:
                os.remove(tmp)
            except OSError:
                pass
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        box = [None if error is None else f"{type(error).__name__}: {error}"]
        dist.broadcast_object_list(box, src=0)  # doubles as the barrier
        if box[0] is not None and error is None:
            raise RuntimeError(f"rank 0 could not write {path}: {box[0]}")
    if error is not None:
        raise error


def multivit_pretrained_keys(state_dict_keys: Iterable[str]) -> List[str]:
    """The tensors a MultiViT backbone is initialised from (reference: MULTIVIT_PRETRAINED_KEYS,
    utils/multimae_keys.py:4-156 = global_tokens, input_adapters.*, encoder.*), in state_dict order."""
    return [k for k in state_dict_keys if k == "global_tokens" or k.startswith("input_adapters.") or k.startswith("encoder.")]


# -- optimizer state <-> torch.optim.AdamW ---------------------------------------------------------------
def _param_order(model):
    decay, no_decay = split_decay(model)
    return decay, no_decay


def optimizer_state_dict(opt: FlatAdamW, model: torch.nn.Module) -> Dict:
    """FlatAdamW state as the state_dict of torch.optim.AdamW([{decayed}, {not decayed}]) over split_decay(model)
    (parameters numbered group by group in model.named_parameters() order)."""
    flat = opt.flat
    off = {name: o for (name, _), o in zip(flat.named, flat.offsets)}
    decay, no_decay = _param_order(model)
    applied = int(opt.step_dev.item()) if opt.step_dev is not None else opt.step_count
    state, groups, idx = {}, [], 0
    for params, wd in ((decay, opt.weight_decay), (no_decay, 0.0)):
        ids = []
        for name, p in params:
            o, n = off[name], p.numel()
            if applied > 0:
                state[idx] = {"step": torch.tensor(float(applied)),
                              "exp_avg": opt.exp_avg[o:o + n].view(p.shape).clone(),
                              "exp_avg_sq": opt.exp_avg_sq[o:o + n].view(p.shape).clone()}
            ids.append(idx)
            idx += 1
        groups.append({"lr": opt.lr, "betas": tuple(opt.betas), "eps": opt.eps, "weight_decay": wd, "amsgrad": False,
                       "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                       "decoupled_weight_decay": True, "params": ids})
    return {"state": state, "param_groups": groups}


def load_optimizer_state_dict(opt: FlatAdamW, model: torch.nn.Module, sd: Dict):
    """Inverse of optimizer_state_dict; accepts what torch.optim.AdamW.state_dict() produced for the same grouping."""
    flat = opt.flat
    off = {name: o for (name, _), o in zip(flat.named, flat.offsets)}
    decay, no_decay = _param_order(model)
    ordered = decay + no_decay
    n_ids = sum(len(g["params"]) for g in sd["param_groups"])
    if n_ids != len(ordered):
        raise ValueError(f"optimizer state holds {n_ids} parameters, the model has {len(ordered)} trainable ones")
    steps = set()
    opt.exp_avg.zero_()
    opt.exp_avg_sq.zero_()
    for idx, (name, p) in enumerate(ordered):
        st = sd["state"].get(idx)
        if st is None:
            continue
        o, n = off[name], p.numel()
        opt.exp_avg[o:o + n].view(p.shape).copy_(st["exp_avg"])
        opt.exp_avg_sq[o:o + n].view(p.shape).copy_(st["exp_avg_sq"])
        steps.add(int(st["step"]))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not the state of a single-schedule AdamW")
    opt.step_count = steps.pop() if steps else 0
    opt.step_dev = None
    g0 = sd["param_groups"][0]
    opt.lr, opt.betas, opt.eps = float(g0["lr"]), tuple(g0["betas"]), float(g0["eps"])
    opt.weight_decay = float(g0["weight_decay"])


# -- files -----------------------------------------------------------------------------------------------
def save_model(output_dir: str, epoch: int, model: torch.nn.Module, optimizer: Optional[FlatAdamW] = None, args=None,
               scaler_state: Optional[Dict] = None) -> str:
    """Write checkpoint-<epoch>.pth in the reference's layout (utils/checkpoint.py:103-129).  bf16 needs no loss scaling,
    so "scaler" holds an empty state unless one is passed in."""
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, "checkpoint-%s.pth" % str(epoch))
    to_save = None
    if _is_main_process():  # the replicas are identical: only the writer gathers 94 M parameters and their moments
        to_save = {"model": OrderedDict((k, v.detach().cpu().clone()) for k, v in model.state_dict().items()), "epoch": epoch,
                   "scaler": dict(scaler_state or {}), "args": args}
        if optimizer is not None:
            osd = optimizer_state_dict(optimizer, model)
            for st in osd["state"].values():
                st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].cpu(), st["exp_avg_sq"].cpu()
            to_save["optimizer"] = osd
    _save_on_master(to_save, path)
    return path


def save_model_for_downstream_tasks(output_dir: str, run_name: str, epoch: int, model: torch.nn.Module) -> str:
    """Encoder-side export multimae_<run>_e<epoch>.pth (utils/checkpoint.py:143-159)."""
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, f"multimae_{run_name}_e{epoch}.pth")
    selected = None
    if _is_main_process():
        sd = model.state_dict()
        selected = OrderedDict((k, sd[k].detach().cpu().clone()) for k in multivit_pretrained_keys(sd.keys()))
    _save_on_master(OrderedDict(model=selected), path)
    return path


def latest_checkpoint(output_dir: str) -> Optional[str]:
    best = -1
    for ckpt in glob.glob(os.path.join(output_dir, "checkpoint-*.pth")):
        t = ckpt.split("-")[-1].split(".")[0]
        if t.isdigit():
            best = max(best, int(t))
    return os.path.join(output_dir, "checkpoint-%d.pth" % best) if best >= 0 else None


def auto_load_model(output_dir: str, model: torch.nn.Module, optimizer: Optional[FlatAdamW] = None, resume: str = "",
                    auto_resume: bool = True, pretrained_weights: Optional[str] = None, trusted: bool = False) -> int:
    """Resume from `resume`, else from the newest checkpoint-<n>.pth in output_dir, else initialise from
    `pretrained_weights` with strict=False (utils/checkpoint.py:162-206).  Returns the epoch to start from.
    Files are read with weights_only=True unless `trusted` (a checkpoint that pickles its argument object needs it)."""
    if auto_resume and not resume:
        resume = latest_checkpoint(output_dir) or ""
    if resume:
        ckpt = torch.load(resume, map_location="cpu", weights_only=not trusted)
        model.load_state_dict(ckpt["model"])
        _after_param_write(optimizer)
        if "optimizer" in ckpt and "epoch" in ckpt and optimizer is not None:
            load_optimizer_state_dict(optimizer, model, ckpt["optimizer"])
            return int(ckpt["epoch"]) + 1
        return 0
    if pretrained_weights is not None:
        ckpt = torch.load(pretrained_weights, map_location="cpu", weights_only=not trusted)
        model.load_state_dict(ckpt["model"], strict=False)
        _after_param_write(optimizer)
    return 0


def _after_param_write(optimizer: Optional[FlatAdamW]):
    if optimizer is not None:
        optimizer.flat.sync_bf16()
