"""Data-parallel step on the GPU: two ranks sharing cuda:0 (gloo collectives over device tensors; bucket hooks fired by
the kernels' in-place gradients, start-up broadcast, 1/world scaling), one rank through torch.distributed's RCCL backend
and one rank through the C ABI's own RCCL driver (the same hooks with the weight-gradient and decoder streams).  (RCCL
does not accept two ranks on one device; the collective backend is the only difference from the multi-GPU run.)"""
import faulthandler
import os
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out):
    # where every thread of this rank stands, should the step still be running after 40 s (read by the parent on a timeout)
    trace = open(f"{out}.rank{rank}.trace", "w")
    faulthandler.dump_traceback_later(40, repeat=False, file=trace)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from tests.test_gpu_model import _build, _losses
    from multimae_amd.train import PretrainStep
    cfg = O.small_cfg()
    model = _build(cfg)
    sd = G.perturbed_state(cfg, seed=7)
    if rank != 0:  # the ranks start from different weights: the step's start-up broadcast must hand out rank 0's
        sd = {k: (v + 0.25 if v.is_floating_point() and not k.endswith("pos_emb") else v) for k, v in sd.items()}
    model.load_state_dict(sd)
    model.cuda()
    step = PretrainStep(model, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, bucket_mb=0.25, clip_grad=0.5)
    assert step.world == 2 and step.buckets.enabled and len(step.buckets.buckets) > 3
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}        # the same batch on both ranks
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    for _ in range(2):
        loss, _, gn2 = step(x, mask_randomness=rnd)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"params": step.flat.params.cpu(), "gn2": gn2.cpu(), "loss": loss.cpu()}, out)
    dist.barrier()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


def _native_worker(rank, out):
    """World size 1 through the C ABI's RCCL driver: no torch.distributed at all."""
    os.environ.update(MMAE_DP_SINGLE_RANK="1", MMAE_COMM="native")
    torch.cuda.set_device(0)
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from tests.test_gpu_model import _build, _losses
    from multimae_amd import _lib
    from multimae_amd.comm import NativeComm
    from multimae_amd.train import PretrainStep
    cfg = O.small_cfg()
    model = _build(cfg)
    model.load_state_dict(G.perturbed_state(cfg, seed=7))
    model.cuda()
    step = PretrainStep(model, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, bucket_mb=0.25, clip_grad=0.5)
    assert isinstance(step.buckets.comm, NativeComm) and step.buckets.enabled and step.world == 1 and len(step.buckets.buckets) > 3
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    for _ in range(2):
        loss, _, gn2 = step(x, mask_randomness=rnd)
    # the collectives themselves, f32 and bf16, and the argument checks
    comm = step.buckets.comm
    t32 = torch.arange(1000, device="cuda", dtype=torch.float32)
    t16 = torch.arange(256, device="cuda", dtype=torch.float32).bfloat16()
    comm.stream.wait_stream(torch.cuda.current_stream())
    comm.all_reduce_(t32); comm.all_reduce_(t16); comm.broadcast_(t32, root=0)
    torch.cuda.current_stream().wait_stream(comm.stream)
    torch.cuda.synchronize()
    assert torch.equal(t32, torch.arange(1000, device="cuda", dtype=torch.float32))
    assert torch.equal(t16.float(), torch.arange(256, device="cuda", dtype=torch.float32))
    lib = _lib.load()
    assert lib.mmae_comm_allreduce_bucket(comm._handle, t32.data_ptr(), 10, 7, None) == -1 and b"dtype" in lib.mmae_last_error()
    assert lib.mmae_comm_broadcast(comm._handle, t32.data_ptr(), 10, 0, 3, None) == -1 and b"root" in lib.mmae_last_error()
    torch.save({"params": step.flat.params.cpu(), "gn2": gn2.cpu(), "loss": loss.cpu()}, out)
    # no host wait inside a step with the native exchange either
    step(x)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        step(x)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    comm.destroy()


def _rccl_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      MMAE_DP_SINGLE_RANK="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dist.barrier()
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from tests.test_gpu_model import _build, _losses
    from multimae_amd.train import PretrainStep
    cfg = O.small_cfg()
    model = _build(cfg)
    model.load_state_dict(G.perturbed_state(cfg, seed=7))
    model.cuda()
    step = PretrainStep(model, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, bucket_mb=0.25, clip_grad=0.5)
    assert step.world == 1 and step.buckets.enabled and len(step.buckets.buckets) > 3
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    for _ in range(2):
        loss, _, gn2 = step(x, mask_randomness=rnd)
    # the max-over-ranks reduction of bench.py
    t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    info = step.buckets.describe()
    assert info["backend"] == "torch-nccl" and info["ranks_seen"] == 1 and info["buckets"] == len(step.buckets.buckets)
    assert info["bytes_per_step"] == 4 * sum(e - s for s, e, _ in step.buckets.buckets) and info["fallback_reason"] is None
    assert len(step.buckets.exposed) == 2  # one (start, end) pair per step around the final wait
    # the same two steps without clipping: the optimizer then runs bucket by bucket, each bucket one behind its all-reduce
    # (on the weight-gradient stream), and must land on the parameters of the plain end-of-step update
    def two_steps(eager):
        m2 = _build(cfg)
        m2.load_state_dict(G.perturbed_state(cfg, seed=7))
        m2.cuda()
        st = PretrainStep(m2, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, bucket_mb=0.25, eager_optimizer=eager)
        assert st.eager is eager and st.buckets.enabled
        for _ in range(2):
            st(x, mask_randomness=rnd)
        torch.cuda.synchronize()
        return st
    eager, plain = two_steps(True), two_steps(False)
    eager_params = eager.flat.params.cpu()
    eager.buckets.timing = []
    eager(x, mask_randomness=rnd)
    torch.cuda.synchronize()
    spans = [(b, a.elapsed_time(c)) for b, a, c in eager.buckets.timing]
    eager.buckets.timing = None
    torch.save({"params": step.flat.params.cpu(), "gn2": gn2.cpu(), "loss": loss.cpu(), "t": t.cpu(),
                "eager": eager_params, "plain": plain.flat.params.cpu(), "spans": spans,
                "n_buckets": len(eager.buckets.buckets)}, out)
    # a further step with the model's own mask draws under torch's synchronisation check: the bucketed all-reduces, their
    # waits and the optimizer's join must not make the host wait for the device
    step(x)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        step(x)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


def _single_rank_reference():
    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from tests.test_gpu_model import _build, _losses
    from multimae_amd.train import PretrainStep
    cfg = O.small_cfg()
    model = _build(cfg)
    model.load_state_dict(G.perturbed_state(cfg, seed=7))
    model.cuda()
    step = PretrainStep(model, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, clip_grad=0.5)
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    for _ in range(2):
        loss, _, gn2 = step(x, mask_randomness=rnd)
    return step, loss, gn2


def test_rccl_bucket_path_with_a_one_rank_group(tmp_path):
    """The RCCL process group (backend "nccl") as bench.py sets it up, with the bucketed all-reduces issued from the
    gradient hooks on both streams: a one-rank group must reproduce the plain step bitwise."""
    out = str(tmp_path / "rccl.pt")
    port = 29900 + os.getpid() % 90
    ctx = mp.spawn(_rccl_worker, args=(port, out), nprocs=1, join=False)
    deadline = time.time() + 90
    while not ctx.join(timeout=5):
        if time.time() > deadline:  # never seen; a hang here must still not silence the rest of the suite
            for proc in ctx.processes:
                if proc.is_alive():
                    proc.kill()
            pytest.fail("the one-rank RCCL step did not finish within 90 s")
    got = torch.load(out)
    step, loss, gn2 = _single_rank_reference()
    assert float(got["t"]) == 1.5
    assert sorted(b for b, _ in got["spans"]) == list(range(got["n_buckets"])) and all(ms >= 0 for _, ms in got["spans"])
    assert torch.equal(got["loss"], loss.cpu())
    assert torch.equal(got["gn2"], gn2.cpu())
    assert torch.equal(got["params"], step.flat.params.cpu())
    # bucket-wise optimizer behind the RCCL group's all-reduces: the same element-wise arithmetic as the one-launch update
    assert torch.equal(got["eager"], got["plain"]), (got["eager"] - got["plain"]).abs().max()


def test_native_rccl_driver_with_a_one_rank_communicator(tmp_path):
    """mmae_comm_* (the C ABI's RCCL driver) behind the gradient buckets: a one-rank communicator must reproduce the plain
    step bitwise, with the all-reduces enqueued on the communicator's own stream."""
    out = str(tmp_path / "native.pt")
    ctx = mp.spawn(_native_worker, args=(out,), nprocs=1, join=False)
    deadline = time.time() + 90
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for proc in ctx.processes:
                if proc.is_alive():
                    proc.kill()
            pytest.fail("the one-rank native RCCL step did not finish within 90 s")
    got = torch.load(out)
    step, loss, gn2 = _single_rank_reference()
    assert torch.equal(got["loss"], loss.cpu()) and torch.equal(got["gn2"], gn2.cpu())
    assert torch.equal(got["params"], step.flat.params.cpu())


def test_two_rank_step_equals_single_rank_on_identical_batches(tmp_path):
    """Both ranks see the same batch and masks and, after the start-up broadcast, hold rank 0's weights, so the averaged
    gradient is the single-rank gradient: (g + g) / 2 == g exactly, and two optimizer steps must land on bitwise identical
    parameters.  The ranks run the step with its default streams (weight gradients and decoders on their own).
    (Earlier stalls of this test were gloo's: its device path waits on the step's streams from high-priority streams, which
    can stop a process's queues on this ROCm stack while the awaited work is pending, see scripts/prio_inversion.py; both
    ranks were then found inside the wait for a gloo collective.  GradBuckets drains the device before each gloo collective
    and finishes it before queueing more work.  A stall is reported with both ranks' stacks.)"""
    out = str(tmp_path / "dp.pt")
    port = 29700 + os.getpid() % 200
    ctx = mp.spawn(_worker, args=(2, port, out), nprocs=2, join=False)
    deadline = time.time() + 90
    while not ctx.join(timeout=5):
        if time.time() > deadline:  # the ranks are ended by process id so that the rest of the suite still runs
            for proc in ctx.processes:
                if proc.is_alive():
                    proc.kill()
            traces = "".join(open(f"{out}.rank{r}.trace").read() for r in range(2) if os.path.exists(f"{out}.rank{r}.trace"))
            pytest.fail("two gloo ranks sharing one GPU did not finish within 90 s; where they stood at 40 s:\n" + traces)
    got = torch.load(out)

    from oracle import multimae_oracle as O
    from tests.golden import common as G
    from tests.test_gpu_model import _build, _losses
    from multimae_amd.train import PretrainStep
    cfg = O.small_cfg()
    model = _build(cfg)
    model.load_state_dict(G.perturbed_state(cfg, seed=7))
    model.cuda()
    step = PretrainStep(model, _losses(cfg), num_encoded_tokens=8, lr=1e-3, weight_decay=0.05, clip_grad=0.5)
    x = {d: t.cuda() for d, t in G.seeded_inputs(cfg, B=4, seed=3).items()}
    rnd = O.draw_mask_randomness(4, [16, 16], 1.0, torch.Generator().manual_seed(2))
    for _ in range(2):
        loss, _, gn2 = step(x, mask_randomness=rnd)
    assert torch.equal(got["loss"], loss.cpu())
    assert torch.equal(got["gn2"], 4 * gn2.cpu())            # norm of the summed gradients: |2g|^2
    assert torch.equal(got["params"], step.flat.params.cpu())
