"""Multi-GPU use of the hot path: one process per GPU, rays sharded, no
data-path collective; the only exchange is the gradient all-reduce of the
4.44 MB parameter blob after backward (what Lightning DDP does for the
reference, nerf_system.py / train.py: `strategy='ddp'`).

`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI between the 8 GPUs of
a node); "gloo" is used by the CPU tests.  The blob is reduced with ONE
collective: per-link ring time for 4.44 MB is ~50 us, so bucketing or overlap
with backward has nothing to win at this size.
"""
import os
from dataclasses import fields
from typing import Optional

import torch
import torch.distributed as dist

from . import utils


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n: int, rank: int, world: int) -> tuple:
    """Contiguous, balanced [begin, end) of `n` items for `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_rays(rays: utils.Rays, rank: int, world: int) -> utils.Rays:
    """This rank's contiguous slice of a flat [R, ...] ray bundle."""
    b, e = shard_bounds(rays.origins.shape[0], rank, world)
    return utils.Rays(*[getattr(rays, f.name)[b:e] for f in fields(rays)])


def _host_staged(t: torch.Tensor, group) -> bool:
    """Device tensor + a host-side backend (gloo): stage collectives through host memory instead of relying on the
    backend's device-tensor support (its all-reduce of the 4.4 MB gradient blob hung in the two-ranks-one-GPU
    smoke run); RCCL ("nccl") works on the device tensors directly."""
    return t.is_cuda and dist.get_backend(group) != "nccl"


def _flat_leaves(module: torch.nn.Module) -> list:
    """MLPs of `module` whose flat blob is a live autograd leaf (MLP.flat_parameter() was called and not released)."""
    seen, out = set(), []
    for m in module.modules():
        f = getattr(m, "_flat", None)
        if hasattr(m, "flat_parameter") and id(m) not in seen and f is not None and f.requires_grad:
            seen.add(id(m))
            out.append(m)
    return out


def allreduce_gradients(module: torch.nn.Module, group=None, average: bool = True, flat: bool = None, force: bool = False) -> None:
    """Sum (or average) the gradients over the group with one all-reduce per blob.
    `flat` (default: `module.config.hip_flat_grads` when the module has a config, i.e. is a Model) selects WHERE the
    gradients live: True = one tensor per MLP (MLP.flat_parameter().grad, reduced in place), False = the nn.Parameters'
    .grad (flattened, reduced, scattered back).  In either mode every rank issues the same collectives: a missing gradient
    contributes zeros.  A module WITHOUT a config (a bare MLP) is in flat mode iff its blob is a live flat leaf -- the state
    flat_parameter() / release_flat_parameter() maintain, identical on every rank that runs the same script; asking such a
    module for the per-parameter path while its flat leaf is live raises (its parameters are views with .grad None: the
    reduction would sum zeros and leave the real gradient unreduced).
    `force`: issue the collective even in a one-rank group (the RCCL smoke test on a one-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    config = getattr(module, "config", None)
    if flat is None:
        flat = bool(getattr(config, "hip_flat_grads", False)) if config is not None else bool(_flat_leaves(module))
    if not flat and config is None and _flat_leaves(module):
        raise RuntimeError("allreduce_gradients(flat=False) on a module whose MLP blob is a live flat leaf (Config.hip_flat_grads / "
                           "MLP.flat_parameter()): the per-parameter .grad tensors are empty in that mode.  Pass flat=True, or call "
                           "MLP.release_flat_parameter() first.")
    if flat:
        seen = set()
        for m in module.modules():
            if not hasattr(m, "flat_parameter") or id(m) in seen:
                continue
            seen.add(id(m))
            blob = m.flat_parameter()
            if blob.grad is None:
                blob.grad = torch.zeros_like(blob)
            g = blob.grad
            if _host_staged(g, group):
                host = g.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                g.copy_(host)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
            if average:
                g /= dist.get_world_size(group)
        return
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return
    cat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    if _host_staged(cat, group):
        host = cat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        cat.copy_(host)
    else:
        dist.all_reduce(cat, op=dist.ReduceOp.SUM, group=group)
    if average:
        cat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        g = cat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None, force: bool = False) -> None:
    """Make every rank start from rank `src`'s parameters (DDP does this at wrap time).  `force`: also in a one-rank group."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    with torch.no_grad():
        for p in module.parameters():
            if _host_staged(p.data, group):
                host = p.data.cpu()
                dist.broadcast(host, src=src, group=group)
                p.data.copy_(host)
            else:
                dist.broadcast(p.data, src=src, group=group)


LAST_IMAGE_COLLECTIVES = 0      # diagnostics: collectives issued by the last render_image_sharded call of this process


def render_image_sharded(render_fn, rays: utils.Rays, config, group=None):
    """models.render_image with the image's rays split across the ranks of
    `group`: every rank renders its contiguous slice of pixels in chunks and the
    per-pixel outputs are all-gathered.  Returns the [H, W, ...] tensors of the
    last level (the `ray_*` visualisation bundles stay local and are dropped)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    height, width = rays.origins.shape[:2]
    n = height * width
    flat = rays.reshape(n, -1)
    mine = shard_rays(flat, rank, world)
    chunks = []
    for i0 in range(0, mine.origins.shape[0], config.render_chunk_size):
        part = utils.Rays(*[getattr(mine, f.name)[i0:i0 + config.render_chunk_size] for f in fields(mine)])
        renderings, _ = render_fn(part)
        chunks.append({k: utils.recursive_detach(v) for k, v in renderings[-1].items() if not k.startswith("ray_")})
    if chunks:
        local = utils.merge_chunks(chunks)
    elif world > 1:
        # a rank without rays (fewer pixels than ranks) still has to join the collective with the same column table: it renders
        # the image's first ray for the keys / dtypes / shapes and contributes zero rows
        probe = utils.Rays(*[getattr(flat, f.name)[:1] for f in fields(flat)])
        renderings, _ = render_fn(probe)
        local = {k: utils.recursive_detach(v)[:0] for k, v in renderings[-1].items() if not k.startswith("ray_")}
    else:
        local = {}
    global LAST_IMAGE_COLLECTIVES
    LAST_IMAGE_COLLECTIVES = 0
    if world == 1:
        return {k: v.reshape((height, width) + v.shape[1:]) for k, v in local.items()}
    sizes = [shard_bounds(n, r, world) for r in range(world)]
    biggest = max(e - b for b, e in sizes)
    # ONE collective per image (it was one all_gather per output key, ~15 per image): the per-ray outputs
    # are packed as byte columns of one [rays, bytes] buffer (float32 colours / distances next to the float64 percentiles, every
    # dtype kept bit for bit), gathered once, and sliced apart again.  Ring all-gather over xGMI is per-link bound: one 124 B/ray
    # message instead of fifteen 4..12 B/ray ones.
    keys = sorted(local)
    cols, width_b = [], 0
    for k in keys:
        v = local[k]
        nb = v[0].numel() * v.element_size() if v.shape[0] else (int(torch.tensor(v.shape[1:]).prod()) if v.dim() > 1 else 1) * v.element_size()
        cols.append((k, width_b, nb, v.dtype, tuple(v.shape[1:])))
        width_b += nb
    pad = torch.zeros((biggest, width_b), dtype=torch.uint8, device=next(iter(local.values())).device)
    for (k, off, nb, _, _) in cols:
        v = local[k]
        pad[:v.shape[0], off:off + nb] = v.contiguous().view(torch.uint8).reshape(v.shape[0], nb)
    if _host_staged(pad, group):
        hpad = pad.cpu()
        hg = [torch.empty_like(hpad) for _ in range(world)]
        dist.all_gather(hg, hpad, group=group)
        gathered = [x.to(pad.device) for x in hg]
    else:
        gathered = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(gathered, pad, group=group)
    LAST_IMAGE_COLLECTIVES = 1
    full = torch.cat([g[:e - b] for g, (b, e) in zip(gathered, sizes)], dim=0)
    out = {}
    for (k, off, nb, dt, tail) in cols:
        # (clone, not contiguous(): a one-row slice IS contiguous at a byte offset that need not be aligned for `dt`)
        out[k] = full[:, off:off + nb].clone().view(dt).reshape((height, width) + tail)
    return out
