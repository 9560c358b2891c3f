"""Benchmark of the MultiMAE ViT-B/16 RGB-D pretraining step (forward, three masked losses, backward with bucketed
gradient all-reduce, AdamW) on synthetic 224^2 tensors resident in HBM.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = global images / second (max step time over ranks).  `roofline` is for the GEMM
kernel family that takes the most time in the step, measured with HIP events around each of its launches in extra,
untimed steps.  `cpu_baseline` (N=1 only) times the CPU restatement of the same step (oracle/, fp32, batch 4)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STEP_GFLOP_PER_IMG = 62.33      # ViT-B RGB-D 224^2, 98 visible tokens, 3 decoders: 3 x forward MACs x 2 (GEMM + attention core)
PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def cpu_baseline():
    """The oracle's CPU step (fp32, batch 4 = the reference's CPU-runnable configuration) on this host's cores."""
    import torch
    from oracle import multimae_oracle as O
    cfg = O.vit_b_rgbd_cfg()
    trainer = O.CpuTrainer(cfg, seed=0)
    g = torch.Generator().manual_seed(1234)
    B = 4
    x = {"rgb": torch.randn(B, 3, 224, 224, generator=g), "depth": torch.rand(B, 1, 224, 224, generator=g)}
    for _ in range(2):
        trainer.step(x, 98, g)
    times, begin = [], time.perf_counter()
    while len(times) < 5 or (time.perf_counter() - begin < 12.0 and len(times) < 60):  # about 12 s of CPU work
        t0 = time.perf_counter()
        trainer.step(x, 98, g)
        times.append(time.perf_counter() - t0)
    n = len(times)
    times.sort()
    med = times[n // 2]
    return {"value": round(B / med, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"ViT-B/16 RGB-D 224^2, 98 visible tokens, 3 decoders, fp32, batch 4: median of {n} steps "
                      "(fwd + 3 losses + bwd + AdamW, about 12 s) after 2 warm-up, oracle/multimae_oracle.py CpuTrainer"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed counter passes of this same command (rocprofv3 cannot run inside
    the process it profiles): profiles/pmc_traffic.json, written by scripts/pmc_table.py from one --pmc FETCH_SIZE and one
    --pmc WRITE_SIZE run, bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 tallies wide reads at half their size).
    Launch-weighted mean over the grids the kernel ran with; (None, None) when the file or the kernel is missing."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, None
    tot = cnt = 0
    for key, v in table.items():
        if kernel in key:
            tot += v["hbm_bytes"] * v["launches"]
            cnt += v["launches"]
    if cnt == 0:
        return None, None
    return int(tot / cnt), "profiles/pmc_traffic.json"


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from multimae_amd import ops
    from multimae_amd.builders import build_losses, build_multimae
    from multimae_amd.train import PretrainStep, init_distributed_mode

    # Rehearsal switches (not used by the measured runs): MMAE_BENCH_BACKEND=gloo and MMAE_BENCH_DEVICE=0 let several ranks
    # share one GPU, which RCCL refuses, to exercise the multi-rank control flow on a one-GPU box.
    rank, world, local = init_distributed_mode(os.environ.get("MMAE_BENCH_BACKEND", "nccl"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device("cuda", int(os.environ.get("MMAE_BENCH_DEVICE", local)))
    torch.cuda.set_device(dev)

    torch.manual_seed(0)  # identical initial weights on every rank (stands for DDP's rank-0 broadcast)
    model = build_multimae().to(dev)
    step = PretrainStep(model, build_losses(), num_encoded_tokens=98, alphas=1.0, lr=1e-4, weight_decay=0.05)
    torch.manual_seed(1234 + rank)  # per-rank data / mask randomness (seed + rank, as the reference)
    B = args.batch
    batch = {"rgb": torch.randn(B, 3, 224, 224, device=dev), "depth": torch.rand(B, 1, 224, 224, device=dev)}

    verbose = os.environ.get("MMAE_BENCH_VERBOSE") == "1"
    if verbose:  # where every thread stands if the run is still going after a minute
        import faulthandler
        faulthandler.dump_traceback_later(60, repeat=False, file=sys.stderr)

    def note(msg):
        if verbose:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    note("model built")
    for i in range(args.warmup):
        loss, _, _ = step(batch)
        note(f"warm-up step {i} enqueued")
    sync()
    note("warm-up done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, task_losses, gn2 = step(batch)
    sync()
    elapsed = time.perf_counter() - t0
    note("timed steps done")
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())
    if not (final_loss == final_loss and abs(final_loss) != float("inf")):
        raise SystemExit(f"non-finite loss {final_loss}")
    ms = elapsed / args.steps * 1e3
    ips = world * B * args.steps / elapsed

    roof = None
    if not args.no_roofline:
        # Two more steps with a HIP-event pair around every GEMM / attention launch (on the stream it is launched on).
        # Every rank runs them (the step holds collectives); rank 0 records.  The weight-gradient stream is switched off
        # here, and so is the decoder stream, so that a kernel's duration is its own, not that of kernels sharing the chip.
        from multimae_amd import functional as Fn
        side, dec = Fn.SIDE_STREAM_DW, Fn.DECODER_STREAMS
        Fn.SIDE_STREAM_DW = Fn.DECODER_STREAMS = False
        if rank == 0:
            ops.PROFILE = []
        for _ in range(2):
            step(batch)
        sync()
        note("per-kernel timing steps done")
        Fn.SIDE_STREAM_DW, Fn.DECODER_STREAMS = side, dec
    if roof is None and not args.no_roofline and rank == 0:
        fam, ker = {}, {}
        for name, flops, e0, e1, label in ops.PROFILE:
            dt = e0.elapsed_time(e1) * 1e-3
            for table, key in ((fam, name), (ker, label)):
                d = table.setdefault(key, [0.0, 0.0, 0])
                d[0] += flops; d[1] += dt; d[2] += 1
        ops.PROFILE = None
        # the dominant kernel = the single kernel (name as rocprofv3 prints it) with the most time per step
        name, (fl, sec, n) = max(((k_, v) for k_, v in ker.items() if v[0] > 0), key=lambda kv: kv[1][1])
        ach = fl / sec / 1e12
        traffic, traffic_src = pmc_traffic(name)
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_step": n // 2,
                "avg_launch_us": round(sec / n * 1e6, 1),
                "kernels": {k_: {"tflops": round(v[0] / v[1] / 1e12, 1) if v[0] > 0 else None, "ms_per_step": round(v[1] / 2 * 1e3, 3),
                                 "launches_per_step": v[2] // 2, "avg_launch_us": round(v[1] / v[2] * 1e6, 1)}
                            for k_, v in sorted(ker.items(), key=lambda kv: -kv[1][1])[:8]},
                "families": {k_: {"tflops": round(v[0] / v[1] / 1e12, 1) if v[0] > 0 else None, "ms_per_step": round(v[1] / 2 * 1e3, 3),
                                  "launches_per_step": v[2] // 2} for k_, v in fam.items()}}

    if rank == 0:
        out = {
            "metric": "pretrain images/sec (node), MultiMAE ViT-B/16 RGB-D 224^2",
            "value": round(ips, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "MultiMAE ViT-B/16 RGB+depth 224^2, 98 visible tokens, 3 decoders (rgb, depth, norm_rgb), "
                                   "bf16 operands / fp32 accumulate, fwd + losses + bwd + grad all-reduce + AdamW",
                       "per_gpu_batch": B, "global_batch": world * B, "num_encoded_tokens": 98, "parallelism": f"dp{world}"},
            "step_gflop_per_image": STEP_GFLOP_PER_IMG,
            "step_mfma_roofline_frac": round(ips * STEP_GFLOP_PER_IMG * 1e9 / (world * PEAK_BF16_TFLOPS * 1e12), 4),
            "final_loss": round(final_loss, 5),
        }
        if roof is not None:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
