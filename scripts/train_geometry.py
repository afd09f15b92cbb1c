"""One-GPU timing of the full training step of configs/refnerf_llff_geometry_losses.gin (BASELINE config 5 shape per
GPU: 2048 rays x 256 samples): clean pass + noisy-ray pass + nine loss terms + backward + Adam.
python scripts/train_geometry.py [rays samples [f32|bf16]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import refnerf_pl_amd  # noqa: F401
from refnerf_pl_amd import configs, models, synthetic, train_utils, utils

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
CHAINS = sys.argv[3] if len(sys.argv) > 3 else "f32"       # 'f32' | 'bf16': arithmetic of the training kernels' MLP chains
dev = torch.device("cuda", 0)
configs.clear_config()
configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_llff_geometry_losses.gin")], [
    f"Model.num_prop_samples = {N}", f"Model.num_nerf_samples = {N}", f"Config.batch_size = {R}",
    f"Config.hip_train_precision = '{CHAINS}'", f"Config.hip_bwd_precision = '{CHAINS}'"])
cfg = configs.Config()
model = models.construct_model(None, cfg).to(dev).train()
model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
rays = utils.rays_from_dict(synthetic.llff_rays(R, seed=1), dev)
batch = utils.Batch(rays=rays, rgb=synthetic.target_rgb(R, seed=7))
opt = torch.optim.Adam(model.parameters(), lr=1e-4)


def step(i):
    opt.zero_grad(set_to_none=True)
    total, losses, stats, aux = train_utils.training_losses(model, batch, rays, cfg, global_step=200000 + i)
    total.backward()
    opt.step()
    return total, losses


for i in range(2):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 8
for i in range(n):
    total, losses = step(i)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / n
extra = cfg.sample_noise_size * cfg.sample_noise_angles
print(f"[{CHAINS} chains] geometry-loss step {R} rays (+{extra} noisy) x {N} samples x 2 levels: {ms:.2f} ms/step = "
      f"{(R + extra) * N * 2 / ms * 1e3:.3e} ray-samples/s; loss {float(total):.5f}; terms {sorted(losses)}")
assert torch.isfinite(total)
