"""One rank of a data-parallel training run (started by tests/test_hip_shards.py; RANK / WORLD_SIZE / MASTER_* in the
environment; gloo backend, the ranks share device 0): `steps` Adam steps on the rank's shard of fixed synthetic batches,
gradients averaged with distributed.allreduce_gradients -- the reference's DDP semantics (train.py:84-88).  Rank 0 writes the
final parameters.  usage: dp_train_worker.py <out.npy> <steps> <rays> <chains> <flat 0|1>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import refnerf_pl_amd  # noqa: E402,F401
from refnerf_pl_amd import configs, distributed, models, synthetic, train_utils, utils  # noqa: E402


def train(out, steps, n_rays, chains, flat, rank, world):
    dev = torch.device("cuda:0")
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")], [
        "Model.num_prop_samples = 48", "Model.num_nerf_samples = 64", f"Config.hip_train_precision = '{chains}'",
        f"Config.hip_bwd_precision = '{chains}'"] + (["Config.hip_flat_grads = True"] if flat else []))
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(dev).train()
    model.nerf_mlp.load_flat_params(synthetic.make_params(seed=4, bias_scale=0.02, sharpen=10.0))
    if world > 1:
        distributed.broadcast_parameters(model)
    params = [model.nerf_mlp.flat_parameter()] if flat else list(model.parameters())
    opt = torch.optim.Adam(params, lr=5e-4, eps=1e-6)
    for it in range(steps):
        rd = synthetic.blender_rays(n_rays, seed=300 + it, center_frac=0.7)
        gt = synthetic.target_rgb(n_rays, seed=400 + it)
        b, e = distributed.shard_bounds(n_rays, rank, world)
        rays = utils.rays_from_dict({k: v[b:e] for k, v in rd.items()}, dev)
        batch = utils.Batch(rays=rays, rgb=gt[b:e])
        opt.zero_grad(set_to_none=True)
        rend, hist = model(rays, 1.0, False)
        total, _, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
        total.backward()
        if world > 1:
            distributed.allreduce_gradients(model)
        opt.step()
        model.nerf_mlp.mark_updated()
    if rank == 0:
        np.save(out, model.nerf_mlp.flat_params().detach().cpu().numpy())


if __name__ == "__main__":
    out, steps, n_rays, chains, flat = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], bool(int(sys.argv[5]))
    rank, world, _ = distributed.init_from_env("gloo")
    train(out, steps, n_rays, chains, flat, rank, world)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
