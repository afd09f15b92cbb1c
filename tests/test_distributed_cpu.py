"""world_size-2 gloo tests (CPU) of the multi-GPU helpers: ray sharding, the
single-collective gradient all-reduce, parameter broadcast and the sharded
image render loop.  The HIP kernels are not involved (no GPU here): render_fn
is a stand-in pure function of the rays."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


GIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "refnerf_blender.gin")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_render(rays):
    o, d = torch.as_tensor(rays.origins), torch.as_tensor(rays.directions)
    rgb = torch.sin(o * 3.0) * 0.5 + d * 0.25
    # (a float64 per-ray output beside the float32 ones, as the level kernels' percentiles: the packed gather keeps every dtype)
    return [{"rgb": rgb * 0.5, "acc": rgb.sum(-1)},
            {"rgb": rgb, "acc": rgb.sum(-1) * 2, "distance_median": rgb.double().sum(-1) / 3.0, "ray_sdist": rgb[:2]}], None


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import refnerf_pl_amd  # noqa: F401
    from refnerf_pl_amd import configs, distributed, models, synthetic, utils
    r, w, _ = distributed.init_from_env("gloo")
    assert (r, w) == (rank, world)
    configs.clear_config()
    configs.parse_config_files_and_bindings([GIN], [])
    cfg = configs.Config()
    torch.manual_seed(100 + rank)                      # different initial weights per rank
    model = models.construct_model(utils.dummy_rays(), cfg)
    distributed.broadcast_parameters(model, src=0)
    blob = model.nerf_mlp.flat_params().clone()
    gathered = [torch.empty_like(blob) for _ in range(world)]
    dist.all_gather(gathered, blob)
    assert all(torch.equal(gathered[0], g) for g in gathered)
    # gradient all-reduce: rank r holds grad = (r+1) * index pattern; average = 1.5 * pattern
    for i, p in enumerate(model.parameters()):
        p.grad = torch.full_like(p, float(rank + 1)) * (1 + i % 3)
    list(model.parameters())[5].grad = None            # a missing gradient counts as zero
    distributed.allreduce_gradients(model, average=True)
    for i, p in enumerate(model.parameters()):
        want = (1.5 if i != 5 else 0.0) * (1 + i % 3)
        assert torch.allclose(p.grad, torch.full_like(p, want)), i
    # flat-gradient mode (Config.hip_flat_grads): ONE blob per MLP, chosen from the configuration -- not from leftover
    # tensor state -- and a rank without a gradient contributes zeros instead of skipping its collective
    cfg.hip_flat_grads = True
    blob = model.nerf_mlp.flat_parameter()
    blob.grad = None if rank == 1 else torch.full_like(blob, 4.0)
    distributed.allreduce_gradients(model, average=True)
    assert torch.allclose(model.nerf_mlp.flat_parameter().grad, torch.full_like(blob, 2.0))
    # ... and back: a stale flat gradient must not shadow the per-parameter gradients
    cfg.hip_flat_grads = False
    for i, p in enumerate(model.parameters()):
        p.grad = torch.full_like(p, float(rank + 1))
    distributed.allreduce_gradients(model, average=True)
    assert all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in model.parameters())
    model.nerf_mlp.release_flat_parameter()
    assert not model.nerf_mlp.flat_params().requires_grad and model.nerf_mlp.flat_params().grad is None
    # a bare MLP has no config: flat mode = "its blob is a live flat leaf" (it used to take the per-parameter
    # path, all-reduce zeros and leave the real flat gradient unreduced)
    mlp = model.nerf_mlp
    blob = mlp.flat_parameter()
    blob.grad = torch.full_like(blob, float(rank + 1))
    distributed.allreduce_gradients(mlp, average=True)
    assert torch.allclose(blob.grad, torch.full_like(blob, 1.5))
    with pytest.raises(RuntimeError):
        distributed.allreduce_gradients(mlp, average=True, flat=False)
    mlp.release_flat_parameter()
    for p in mlp.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    distributed.allreduce_gradients(mlp, average=True)
    assert all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in mlp.parameters())
    # ray sharding covers every ray exactly once, in order
    rd = synthetic.blender_rays(37, seed=3)
    rays = utils.rays_from_dict(rd)
    b, e = distributed.shard_bounds(37, rank, world)
    mine = distributed.shard_rays(rays, rank, world)
    assert mine.origins.shape[0] == e - b
    cnt = torch.tensor([e - b])
    dist.all_reduce(cnt)
    assert int(cnt) == 37
    # sharded image render == single-process render
    img = utils.rays_from_dict({k: np.asarray(v)[:35].reshape(7, 5, -1) for k, v in rd.items()})
    cfg.render_chunk_size = 4
    out = distributed.render_image_sharded(_fake_render, img, cfg)
    ref, _ = _fake_render(utils.rays_from_dict({k: np.asarray(v)[:35] for k, v in rd.items()}))
    assert set(out) == {"rgb", "acc", "distance_median"}
    assert torch.equal(out["rgb"].reshape(-1, 3), ref[-1]["rgb"]) and out["rgb"].shape == (7, 5, 3)
    assert torch.equal(out["acc"].reshape(-1), ref[-1]["acc"])
    assert out["distance_median"].dtype == torch.float64 and torch.equal(out["distance_median"].reshape(-1), ref[-1]["distance_median"])
    assert distributed.LAST_IMAGE_COLLECTIVES == 1       # ONE all_gather per image, whatever the number of outputs
    # a 1 x 1 image on two ranks: one rank has no ray at all, yet joins the collective with the same column table; and a single
    # gathered row must come apart at byte offsets that are not float64-aligned
    one = utils.rays_from_dict({k: np.asarray(v)[:1].reshape(1, 1, -1) for k, v in rd.items()})
    out1 = distributed.render_image_sharded(_fake_render, one, cfg)
    ref1, _ = _fake_render(utils.rays_from_dict({k: np.asarray(v)[:1] for k, v in rd.items()}))
    assert out1["rgb"].shape == (1, 1, 3) and torch.equal(out1["rgb"].reshape(-1, 3), ref1[-1]["rgb"])
    assert out1["distance_median"].dtype == torch.float64 and torch.equal(out1["distance_median"].reshape(-1), ref1[-1]["distance_median"])
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_two_rank_gloo_helpers(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_shard_bounds_properties():
    from refnerf_pl_amd import distributed
    for n in (0, 1, 7, 8, 4096, 640000):
        for world in (1, 2, 3, 8):
            spans = [distributed.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_noop():
    from refnerf_pl_amd import configs, distributed, models, utils
    configs.clear_config()
    configs.parse_config_files_and_bindings([GIN], [])
    model = models.construct_model(utils.dummy_rays(), configs.Config())
    configs.clear_config()
    distributed.allreduce_gradients(model)        # not initialised: no-op
    distributed.broadcast_parameters(model)


def test_bench_launcher_starts_ranks_and_fails_loudly_without_gpus():
    """`bench.py --gpus N` (N > 1, outside torchrun) is a PARENT that starts N rank processes before touching a GPU.
    On this CPU-only box: (a) it refuses N ranks on fewer devices (exit 2, nothing started); (b) with the share-GPU
    smoke override it does start the two ranks, each of which fails loudly for lack of a device (no CPU fallback),
    and the parent reports a non-zero exit."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "REFNERF_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "only" in r.stderr and r.stdout == ""
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=dict(env, REFNERF_BENCH_SHARE_GPU="1", REFNERF_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "rank exit codes" in r.stderr and r.stdout == ""
