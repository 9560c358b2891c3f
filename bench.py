"""Benchmark of the MultiMAE pretraining step (forward, masked losses, backward with bucketed gradient all-reduce, AdamW)
on synthetic tensors resident in HBM.  Default workload: ViT-B/16 RGB-D 224^2, 98 visible tokens, 3 decoders, 256 images
per GPU (the configuration the headline metric is quoted on); --config selects the other pretraining configurations.

  python bench.py --gpus 1 --steps 20 --warmup 5 [--config vitb_rgbd | vitb_rgbd_semseg | vitl384]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = global images / second (max step time over ranks).  `roofline` is for the GEMM
kernel that takes the most time in the step, measured with HIP events around each of its launches in extra, untimed
steps; `roofline.hbm_kernels` gives the same for the bandwidth-bound kernels (masking, losses, LayerNorm, AdamW) against
the HBM peak.  `cpu_baseline` (N=1 only) times the CPU restatement of the same step (oracle/, fp32, small batch)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0

# The pretraining configurations.  gflop = algorithmic FLOPs of one step per image: 3 x forward MACs x 2 over the matrix
# products and the attention cores as the reference computes them (patch projection counted for every token).
CONFIGS = {
    # pretrain_argparser.py:13-64 defaults / cfgs/pretrain/multimae-b_98_rgb-depth_400e.yaml of the reference
    "vitb_rgbd": dict(model="pretrain_multimae_base", in_domains=("rgb", "depth"), out_domains=("rgb", "depth"), image=224,
                      k=98, batch=256, gflop=62.33, fp32_out=(), cpu_batch=4,
                      name="MultiMAE ViT-B/16 RGB+depth 224^2, 98 visible tokens, 3 decoders (rgb, depth, norm_rgb)"),
    # cfgs/pretrain/multimae-b_98_rgb+-depth-semseg_1600e.yaml:13-28, run_pretraining_multimae.py:72-85
    "vitb_rgbd_semseg": dict(model="pretrain_multimae_base", in_domains=("rgb", "depth", "semseg"),
                             out_domains=("rgb", "depth", "semseg"), image=224, k=98, batch=256, gflop=67.01,
                             fp32_out=("semseg",), cpu_batch=4,
                             name="MultiMAE ViT-B/16 RGB+depth+semseg(133 classes, stride 4) 224^2, 98 of 588 tokens visible, "
                                  "4 decoders (rgb, depth, semseg in fp32, norm_rgb)"),
    # multimae/multimae.py:461-478 (ViT-L: 1024 wide, 24 layers, 16 heads) at 384^2; 288 of 1152 tokens visible (the same
    # quarter as 98 of 392); the reference's per-GPU batch of 256 fits (88 GB of the 288): 64 / 128 / 256 images per step
    # measured 1,086 / 1,221 / 1,281 images/s (the optimizer's 9 GB pass and the tile rounds amortise)
    "vitl384": dict(model="pretrain_multimae_large", in_domains=("rgb", "depth"), out_domains=("rgb", "depth"), image=384,
                    k=288, batch=256, gflop=586.3, fp32_out=(), cpu_batch=1,
                    name="MultiMAE ViT-L/16 RGB+depth 384^2, 288 visible tokens, 3 decoders (rgb, depth, norm_rgb)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="vitb_rgbd")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--comm", choices=("torch", "native"), default=None,
                    help="gradient exchange for N > 1: torch.distributed's RCCL group (default) or the C ABI's own RCCL "
                         "communicator (mmae_comm_*); sets MMAE_COMM")
    ap.add_argument("--comm-cus", type=int, default=0,
                    help="CUs left to the gradient exchange: caps RCCL at that many channels (NCCL_MAX_NCHANNELS, unless set) "
                         "and plans the one-workgroup-per-CU GEMM kernels for the remaining CUs (MMAE_CU_BUDGET); 0 = off")
    ap.add_argument("--no-standardize-depth", action="store_true",
                    help="leave the truncated depth standardisation (the reference's default, on) out of the step")
    return ap.parse_args()


def synthetic_batch(conf, B, device, generator=None):
    """rgb ~ N(0, 1) (normalised pixels), depth ~ U[0, 1), semseg = uniform class ids at a quarter of the resolution."""
    import torch
    S = conf["image"]
    kw = dict(device=device, generator=generator)
    x = {"rgb": torch.randn(B, 3, S, S, **kw), "depth": torch.rand(B, 1, S, S, **kw)}
    if "semseg" in conf["in_domains"]:
        x["semseg"] = torch.randint(0, 133, (B, S // 4, S // 4), **kw)
    return x


def oracle_cfg(conf):
    from oracle import multimae_oracle as O
    if "semseg" in conf["in_domains"]:
        return O.vit_b_rgbd_semseg_cfg(extra_norm_pix=True, image=conf["image"])
    if conf["model"] == "pretrain_multimae_large":
        out = {"rgb": (3, "rgb"), "depth": (1, "depth"), "norm_rgb": (3, "rgb")}
        return O.make_cfg({"rgb": 3, "depth": 1}, out, dim=1024, depth=24, heads=16, image=conf["image"])
    return O.vit_b_rgbd_cfg(image=conf["image"])


def cpu_baseline(conf):
    """The oracle's CPU step (fp32, a small batch of the same configuration) on this host's cores."""
    import torch
    from oracle import multimae_oracle as O
    trainer = O.CpuTrainer(oracle_cfg(conf), seed=0)
    g = torch.Generator().manual_seed(1234)
    B, k = conf["cpu_batch"], conf["k"]
    x = synthetic_batch(conf, B, "cpu", g)
    warm = 2 if conf["gflop"] < 100 else 1
    for _ in range(warm):
        trainer.step(x, k, g)
    times, begin = [], time.perf_counter()
    while len(times) < 5 or (time.perf_counter() - begin < 12.0 and len(times) < 60):  # about 12 s of CPU work
        t0 = time.perf_counter()
        trainer.step(x, k, g)
        times.append(time.perf_counter() - t0)
    n = len(times)
    times.sort()
    med = times[n // 2]
    return {"value": round(B / med, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{conf['name']}, fp32, batch {B}: median of {n} steps (fwd + losses + bwd + AdamW, about 12 s) "
                      f"after {warm} warm-up, oracle/multimae_oracle.py CpuTrainer"}


def pmc_traffic(kernel, config="vitb_rgbd"):
    """HBM bytes per launch of `kernel` from the committed counter passes of this same command (rocprofv3 cannot run inside
    the process it profiles): profiles/pmc_traffic.json, written by scripts/pmc_table.py from one --pmc FETCH_SIZE and one
    --pmc WRITE_SIZE run, bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 tallies wide reads at half their size).
    Launch-weighted mean over the grids the kernel ran with; (None, None) when the file or the kernel is missing."""
    rel = "profiles/pmc_traffic.json" if config == "vitb_rgbd" else f"profiles/pmc_traffic_{config}.json"
    path = os.path.join(ROOT, rel)
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
    return int(tot / cnt), rel


def main():
    args = parse()
    conf = CONFIGS[args.config]
    if args.comm is not None:
        os.environ["MMAE_COMM"] = "native" if args.comm == "native" else "torch"
    if args.comm_cus > 0:  # before the communicator and the kernel library come up: both read the environment once
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(args.comm_cus))
        os.environ["MMAE_CU_BUDGET"] = str(-args.comm_cus)  # negative: that many fewer than the device has
    # Only the result line goes to stdout.  Libraries write there on their own when a process group starts (RCCL's version
    # banner, gloo's connection notes): for the duration of the run file descriptor 1 points at stderr, and rank 0 puts it
    # back just for the JSON line.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from multimae_amd import functional as Fn
    from multimae_amd import ops
    from multimae_amd.builders import build_losses, build_multimae
    from multimae_amd.train import PretrainStep, init_distributed_mode

    # Rehearsal switches (not used by the measured runs): MMAE_BENCH_BACKEND=gloo and MMAE_BENCH_DEVICE=0 let several ranks
    # share one GPU, which RCCL refuses, to exercise the multi-rank control flow on a one-GPU box.
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("MMAE_BENCH_DEVICE", local)))
    torch.cuda.set_device(dev)
    # The step's own streams (weight gradients, decoders) are chosen before the process group exists: the choice times
    # idle kernels on candidate streams, and must not do so beside a resident collective kernel.
    Fn.prepare_streams(dev, len(conf["out_domains"]) + 1)
    rank, world, local = init_distributed_mode(os.environ.get("MMAE_BENCH_BACKEND", "nccl"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    torch.manual_seed(0 + rank)  # seed + rank before the model is built, as the reference; rank 0's weights are broadcast
    model = build_multimae(conf["model"], in_domains=conf["in_domains"], out_domains=conf["out_domains"],
                           image_size=conf["image"]).to(dev)
    # standardize_depth: the reference's default (pretrain_argparser.py:20, applied at run_pretraining_multimae.py:448-459)
    step = PretrainStep(model, build_losses(conf["out_domains"]), num_encoded_tokens=conf["k"], alphas=1.0, lr=1e-4,
                        weight_decay=0.05, fp32_output_adapters=conf["fp32_out"],
                        standardize_depth=not args.no_standardize_depth)
    comm_info = step.buckets.describe()
    comm_info["requested"] = os.environ.get("MMAE_COMM", "torch")
    comm_info["eager_optimizer"] = bool(step.eager)
    print(f"[bench rank {rank}] gradient exchange: {comm_info['backend']} over {comm_info['ranks_seen']} rank(s), "
          f"{comm_info['buckets']} buckets, {comm_info['bytes_per_step'] / 1e6:.1f} MB per step, optimizer "
          f"{'bucket by bucket' if step.eager else 'at the end of the step'}", file=sys.stderr, flush=True)
    if comm_info["ranks_seen"] != world:
        raise SystemExit(f"rank {rank}: the communicator spans {comm_info['ranks_seen']} ranks, the launcher started {world}")
    torch.manual_seed(1234 + rank)  # per-rank data / mask randomness
    B = args.batch or conf["batch"]
    batch = synthetic_batch(conf, B, dev)

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
    step.buckets.exposed.clear()
    # one event per step on the step's stream (no host wait): the spread of the per-step times, beside the mean that `value` uses
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss, task_losses, gn2 = step(batch)
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    # the part of the gradient exchange that backward did not hide: how long the step's stream sat in the final wait
    exposed = [a.elapsed_time(b) for a, b in step.buckets.exposed]
    comm_info["exposed_ms_per_step"] = round(sum(exposed) / len(exposed), 3) if exposed else 0.0
    note("timed steps done")
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())
    if not (final_loss == final_loss and abs(final_loss) != float("inf")):
        raise SystemExit(f"non-finite loss {final_loss}")
    ms = elapsed / args.steps * 1e3
    ips = world * B * args.steps / elapsed
    peak_mem = torch.cuda.max_memory_allocated(dev)

    roof = None
    if not args.no_roofline:
        # Two more steps with a HIP-event pair around every GEMM / attention / bandwidth-kernel launch (on the stream it is
        # launched on).  Every rank runs them (the step holds collectives); rank 0 records.  The weight-gradient stream is
        # switched off here, and so is the decoder stream, so that a kernel's duration is its own, not that of kernels
        # sharing the chip.
        side, dec = Fn.SIDE_STREAM_DW, Fn.DECODER_STREAMS
        Fn.SIDE_STREAM_DW = Fn.DECODER_STREAMS = False
        if rank == 0:
            ops.PROFILE = []
        step.buckets.timing = []  # an event pair around every bucket's collective, on the stream it is issued from
        for _ in range(2):
            step(batch)
        sync()
        spans = [(b, a.elapsed_time(c)) for b, a, c in step.buckets.timing]
        step.buckets.timing = None
        comm_info["allreduce_ms_per_step"] = round(sum(ms_ for _, ms_ in spans) / 2, 3)
        comm_info["allreduce_ms_per_bucket"] = [round(sum(ms_ for b, ms_ in spans if b == i) / 2, 3)
                                                for i in range(comm_info["buckets"])]
        if comm_info["allreduce_ms_per_step"] > 0:
            comm_info["allreduce_bus_gbps"] = round(2 * (world - 1) / world * comm_info["bytes_per_step"]
                                                    / comm_info["allreduce_ms_per_step"] / 1e6, 1)
        note("per-kernel timing steps done")
        Fn.SIDE_STREAM_DW, Fn.DECODER_STREAMS = side, dec
    if not args.no_roofline and rank == 0:
        fam, ker, hbm = {}, {}, {}
        for name, flops, e0, e1, label, nbytes in ops.PROFILE:
            dt = e0.elapsed_time(e1) * 1e-3
            if nbytes is not None:
                d = hbm.setdefault(label, [0.0, 0.0, 0])
                d[0] += nbytes; d[1] += dt; d[2] += 1
                continue
            for table, key in ((fam, name), (ker, label)):
                d = table.setdefault(key, [0.0, 0.0, 0])
                d[0] += flops; d[1] += dt; d[2] += 1
        ops.PROFILE = None
        # the dominant kernel = the single kernel (name as rocprofv3 prints it) with the most time per step
        name, (fl, sec, n) = max(((k_, v) for k_, v in ker.items() if v[0] > 0), key=lambda kv: kv[1][1])
        ach = fl / sec / 1e12
        traffic, traffic_src = pmc_traffic(name, args.config)

        def hbm_row(label, v):
            gbps = v[0] / v[1] / 1e9
            counter, _ = pmc_traffic(label.split(" ")[0], args.config)
            return {"bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": round(gbps / PEAK_HBM_GBPS, 4), "algorithmic_bytes_per_launch": int(v[0] / v[2]),
                    "traffic": counter, "launches_per_step": v[2] // 2, "avg_launch_us": round(v[1] / v[2] * 1e6, 1),
                    "ms_per_step": round(v[1] / 2 * 1e3, 3)}
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_step": n // 2,
                "avg_launch_us": round(sec / n * 1e6, 1),
                "kernels": {k_: {"tflops": round(v[0] / v[1] / 1e12, 1) if v[0] > 0 else None, "ms_per_step": round(v[1] / 2 * 1e3, 3),
                                 "launches_per_step": v[2] // 2, "avg_launch_us": round(v[1] / v[2] * 1e6, 1)}
                            for k_, v in sorted(ker.items(), key=lambda kv: -kv[1][1])[:8]},
                "families": {k_: {"tflops": round(v[0] / v[1] / 1e12, 1) if v[0] > 0 else None, "ms_per_step": round(v[1] / 2 * 1e3, 3),
                                  "launches_per_step": v[2] // 2} for k_, v in fam.items()},
                # the bandwidth-bound kernels against the HBM peak: algorithmic bytes (what the kernel must read and
                # write once) / HIP-event time; `traffic` = counter bytes per launch from the committed passes, when present
                "hbm_kernels": {k_: hbm_row(k_, v) for k_, v in sorted(hbm.items(), key=lambda kv: -kv[1][1])}}

    if rank == 0:
        out = {
            "metric": "pretrain images/sec (node), " + conf["name"].split(",")[0],
            "value": round(ips, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "ms_per_step_median": round(per_step[len(per_step) // 2], 3),
            "ms_per_step_p10": round(per_step[len(per_step) // 10], 3), "ms_per_step_p90": round(per_step[(9 * len(per_step)) // 10], 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": conf["name"] + ", bf16 operands / fp32 accumulate, fwd + losses + bwd + grad all-reduce + AdamW",
                       "name": args.config, "per_gpu_batch": B, "global_batch": world * B, "num_encoded_tokens": conf["k"],
                       "parallelism": f"dp{world}"},
            "step_gflop_per_image": conf["gflop"],
            "step_mfma_roofline_frac": round(ips * conf["gflop"] * 1e9 / (world * PEAK_BF16_TFLOPS * 1e12), 4),
            "final_loss": round(final_loss, 5),
            "peak_device_memory_gb": round(peak_mem / 1e9, 2),
        }
        if roof is not None:
            out["roofline"] = roof
        if world > 1 or comm_info["enabled"]:
            from multimae_amd import _lib
            comm_info["planned_cus"] = int(_lib.load().mmae_get_cu_budget())  # CUs the one-workgroup-per-CU kernels plan for
            comm_info["nccl_max_nchannels"] = os.environ.get("NCCL_MAX_NCHANNELS")
            out["comm"] = comm_info
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(conf)
        sys.stdout.flush()
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
