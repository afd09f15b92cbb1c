#!/usr/bin/env python3
"""bench.py -- Ref-NeRF rendering inner loop on MI355X.

One "step" = one ``Model.__call__`` (eval forward, compute_extras=True) over a
synthetic Blender-style batch of 4096 rays x 128 samples x 2 levels
(BASELINE.json configs[1]) through the fused HIP path.  Prints ONE JSON line.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: rays shard by rank (each rank renders its own 4096-ray tile, weights
replicated, no data-path collective) -> weak scaling; RCCL is used only for the
timing barrier / max-over-ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_SAMPLE = 2211840            # MLP contractions only
TRAIN_FLOP_PER_SAMPLE = 7651840      # fwd + density-normal VJP + backward
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # MI355X dense MFMA peaks (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--precision", default="bf16", choices=["f32", "bf16"],
                    help="arithmetic of the MLP contractions for the headline value; the other mode is reported alongside")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-image", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    return ap.parse_args()


def build_model(args, dev):
    import refnerf_pl_amd  # noqa: F401
    from refnerf_pl_amd import configs, models, synthetic
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")], [
        f"Model.num_prop_samples = {args.samples}", f"Model.num_nerf_samples = {args.samples}",
        f"Config.batch_size = {args.rays}", f"Config.hip_precision = '{args.precision}'"])
    cfg = configs.Config()
    model = models.construct_model(None, cfg).to(dev)
    model.eval()
    blob = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0)
    model.nerf_mlp.load_flat_params(blob)
    return model, cfg, blob


def cpu_baseline(blob, args, target_seconds):
    """Time the CPU oracle (a from-scratch C port, OpenMP over rays, all host
    cores) on a bounded sample of the same workload."""
    from oracle import oracle as O
    from refnerf_pl_amd import synthetic
    cores = os.cpu_count() or 1
    # two calibration rounds (64 rays, then ~2 s worth) so that the timed sample lands in the 10-30 s window
    # whatever the core count: the first call also pays for thread start-up and page faults
    n_probe, rate = 64, None
    for _ in range(3):
        probe = synthetic.blender_rays(n_probe, seed=1, center_frac=0.5)
        t0 = time.time()
        O.model_forward(blob, probe, num_prop_samples=args.samples, num_nerf_samples=args.samples, n_threads=cores, history=True)
        rate = n_probe * args.samples * 2 / max(time.time() - t0, 1e-6)
        n_probe = max(64, min(4096, int(rate * 2.0 / (args.samples * 2)) // 64 * 64))
    n_rays = int(min(16 * args.rays, max(64, rate * target_seconds / (args.samples * 2))))
    n_rays = max(64, (n_rays // 64) * 64)
    rays = synthetic.blender_rays(n_rays, seed=1, center_frac=0.5)
    t0 = time.time()
    O.model_forward(blob, rays, num_prop_samples=args.samples, num_nerf_samples=args.samples, n_threads=cores, history=True)
    dt = time.time() - t0
    return {"value": n_rays * args.samples * 2 / dt, "unit": "ray-samples/s", "cores": cores, "kind": "port",
            "sample": f"{n_rays} rays (same generator as the {args.rays}-ray batch) x {args.samples} samples x 2 levels, "
                      f"eval forward, fp32, {dt:.1f} s"}


def train_step_bench(args, model, cfg, rays, rank, world, dev, dist, sync, chains="f32"):
    """Secondary figure: one full training step (training forward with density-gradient
    normals + saved layer inputs, the three Ref-NeRF losses, HIP backward + weight-gradient GEMM, gradient
    all-reduce over the ranks, Adam step) on the same batch; fp32 MFMA chains, the weight-gradient GEMM on
    split-bf16 MFMA at fp32 accuracy (Config.hip_wgrad_mode)."""
    from refnerf_pl_amd import distributed, synthetic, train_utils, utils
    model.train()
    cfg.hip_train_precision = cfg.hip_bwd_precision = chains      # 'f32' (parity mode) | 'bf16' (bf16 MFMA chains)
    gt = synthetic.target_rgb(args.rays, seed=7 + rank)
    batch = utils.Batch(rays=rays, rgb=gt)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        renderings, history = model(rays, 1.0, False)
        total, terms, _ = train_utils.compute_losses(model, batch, rays, renderings, history, cfg)
        total.backward()
        distributed.allreduce_gradients(model)
        opt.step()
        return total

    n = max(2, min(10, args.steps // 3))
    step()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        loss = step()
    sync()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    model.eval()
    cfg.hip_train_precision = cfg.hip_bwd_precision = "f32"
    assert torch.isfinite(loss.detach()).all()
    rate = world * args.rays * args.samples * 2 * n / el
    tf = rate / world * TRAIN_FLOP_PER_SAMPLE / 1e12
    return {"value": rate, "unit": "ray-samples/s (fwd+bwd+Adam)", "ms_per_step": 1e3 * el / n, "steps": n,
            "dtype": chains, "wgrad": getattr(cfg, "hip_wgrad_mode", "bf16x3"), "loss": float(loss.detach()),
            "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS[chains], "unit": "TFLOP/s",
                         "frac": tf / PEAK_TFLOPS[chains],
                         "note": "whole step incl. losses, optimiser and weight re-pack; algorithmic 7,651,840 FLOP/ray-sample; "
                                 "the weight-gradient GEMM runs on split-bf16 MFMA in both modes and is HBM-bound"}}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # REFNERF_BENCH_BACKEND=gloo + REFNERF_BENCH_SHARE_GPU=1: smoke-test the N>1 code path on a 1-GPU box
        backend = os.environ.get("REFNERF_BENCH_BACKEND", "nccl")
        if os.environ.get("REFNERF_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from refnerf_pl_amd import _hip, synthetic, utils
    _hip.require_device()
    model, cfg, blob = build_model(args, dev)
    # ray-tile data parallel: rank r renders its own tile of the (virtual) image
    rays = utils.rays_from_dict(synthetic.blender_rays(args.rays, seed=1 + rank, center_frac=0.5), dev)

    def step():
        with torch.no_grad():
            return model(rays, 1.0, True)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, n_warm):
        """n_warm untimed steps, then exactly n_steps bracketed by barrier + synchronize on
        both sides; returns (elapsed s [max over ranks], kernel ms total, launches, last output)."""
        for _ in range(n_warm):
            step()
        sync()
        _hip.set_timing(True)      # HIP event pairs on the kernel's own stream, inside the library
        t0 = time.perf_counter()
        for _ in range(n_steps):
            out = step()
        sync()
        el = time.perf_counter() - t0
        kern_ms, launches = _hip.get_timing()
        _hip.set_timing(False)
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, kern_ms, launches, out

    def roofline(prec, kern_ms, launches):
        avg_ms = kern_ms / launches
        flop_per_launch = args.rays * args.samples * FLOP_PER_SAMPLE
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12
        kernel = "rn::level_fwd_f32" if prec == "f32" else "rn::level_fwd_bf16"
        traffic = None
        try:   # PMC numbers cannot be collected inside this process: taken from the committed rocprofv3 passes
            prof = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[kernel]
            if args.rays == 4096 and args.samples == 128:
                traffic = prof["bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        return {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[prec], "unit": "TFLOP/s",
                "frac": achieved / PEAK_TFLOPS[prec], "traffic": traffic, "kernel": kernel,
                "avg_launch_ms": avg_ms, "launches": launches, "flop_per_launch": flop_per_launch}

    elapsed, kern_ms, launches, out = timed(args.steps, args.warmup)
    rgb = out[0][-1]["rgb"]
    assert torch.isfinite(rgb).all()

    samples_per_step = args.rays * args.samples * 2
    value = world * samples_per_step * args.steps / elapsed
    line = {
        "metric": "ray-samples/s (4096 rays x 128 samples x 2 levels, Ref-NeRF Blender, eval forward)",
        "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"blender_refnerf.gin, {args.rays} rays x {args.samples} samples, 2-level, "
                               "Model.__call__ eval forward with compute_extras + full ray_history",
                   "rays_per_gpu": args.rays, "parallelism": f"ray-tile dp{world}"},
    }
    if launches:
        line["roofline"] = roofline(args.precision, kern_ms, launches)
    if rank == 0 and world == 1:
        # the other arithmetic mode on the same batch (f32 = exact-fp32 MFMA, the strict parity mode)
        other = "f32" if args.precision == "bf16" else "bf16"
        cfg.hip_precision = other
        n2 = max(3, args.steps // 5)
        el2, k2, l2, out2 = timed(n2, 1)
        line[other + "_mode"] = {"value": samples_per_step * n2 / el2, "unit": "ray-samples/s",
                                 "ms_per_step": 1e3 * el2 / n2, "dtype": other, "roofline": roofline(other, k2, l2)}
        line["mode_agreement"] = {"rgb_linf_bf16_vs_f32": float((out[0][-1]["rgb"] - out2[0][-1]["rgb"]).abs().max()),
                                  "note": "rendered RGB of the two modes on this batch; parity of each mode vs the "
                                          "reference's golden vectors is asserted in tests/test_hip_parity.py"}
        cfg.hip_precision = args.precision
    if rank == 0 and world == 1 and not args.no_image:
        # full-image render ms: 800x800 Blender view, 157 chunks of 4096 rays (models.render_image)
        from refnerf_pl_amd import models
        # rays of the whole view are cast on the device (refnerf_pixels_to_rays), inside the timed region
        from refnerf_pl_amd import camera_utils
        c2w, focal = synthetic.blender_camera(seed=1)
        # two renders: the first one also pays for the one-off growth of torch's caching allocator (157 chunks of
        # outputs) and the first launch of the ray-casting kernel; the headline is the steady state
        for tag in ("full_image_render_first_ms", "full_image_render_ms"):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                img = camera_utils.cast_pinhole_rays(c2w.astype(np.float32), 800, 800, focal, 2.0, 6.0, device=dev)
                rendering = models.render_image(lambda r: model(r, 1.0, True), img, cfg, verbose=False, device=dev)
            torch.cuda.synchronize()
            line[tag] = 1e3 * (time.perf_counter() - t0)
        assert rendering["rgb"].shape == (800, 800, 3)
        # 1008x756 LLFF-style view (NDC rays, near 0 / far 1), same chunked loop
        del img, rendering
        lr = synthetic.llff_rays(0, seed=1, full_image=True)
        img = utils.rays_from_dict({k: v.reshape(756, 1008, -1) for k, v in lr.items()}, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            rendering = models.render_image(lambda r: model(r, 1.0, True), img, cfg, verbose=False, device=dev)
        torch.cuda.synchronize()
        line["llff_image_render_ms"] = 1e3 * (time.perf_counter() - t0)
        assert rendering["rgb"].shape == (756, 1008, 3)
    if not args.no_train:
        line["train_step"] = train_step_bench(args, model, cfg, rays, rank, world, dev, dist, sync, "f32")
        line["train_step_bf16"] = train_step_bench(args, model, cfg, rays, rank, world, dev, dist, sync, "bf16")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(blob, args, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
