"""Training step in the arithmetic modes of the training kernels (Config.hip_train_precision / hip_bwd_precision):
step time, loss and gradient agreement with the all-f32 mode.  python scripts/ab_trainprec.py [rays samples]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import refnerf_pl_amd  # noqa: F401
from refnerf_pl_amd import configs, models, synthetic, train_utils, utils

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
out = {}
for fwd, bwd in (("f32", "f32"), ("f32", "bf16"), ("bf16", "bf16"), ("f32", "f32"), ("bf16", "bf16")):
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")], [
        f"Model.num_prop_samples = {N}", f"Model.num_nerf_samples = {N}", f"Config.hip_train_precision = '{fwd}'",
        f"Config.hip_bwd_precision = '{bwd}'"])
    cfg = configs.Config()
    model = models.construct_model(None, cfg).to(dev).train()
    model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
    rays = utils.rays_from_dict(synthetic.blender_rays(R, seed=1, center_frac=0.5), dev)
    batch = utils.Batch(rays=rays, rgb=synthetic.target_rgb(R, seed=7))

    def step():
        model.zero_grad(set_to_none=True)
        renderings, history = model(rays, 1.0, False)
        total, _, _ = train_utils.compute_losses(model, batch, rays, renderings, history, cfg)
        total.backward()
        return total, renderings
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    for _ in range(n):
        total, rend = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    g = torch.cat([p.grad.flatten() for p in model.nerf_mlp.ordered_parameters()]).double().cpu()
    out[(fwd, bwd)] = (g, rend[1]["rgb"].detach().cpu())
    print(f"fwd {fwd:4s} bwd {bwd:4s} {R}x{N}: {ms:.2f} ms per fwd+bwd   loss {float(total):.7f}  |g| {float(g.norm()):.6e}", flush=True)
a, rgb_a = out[("f32", "f32")]
for k, (b, rgb_b) in out.items():
    print(k, "grad rel L2 vs f32/f32:", float((a - b).norm() / a.norm()), " rgb L-inf:", float((rgb_a - rgb_b).abs().max()))
