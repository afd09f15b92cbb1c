"""End-to-end training with the three chain modes from the same initialisation: python scripts/train_curves.py [steps] [rays] [samples]
-> one JSON object (data-loss curve every 25 steps, final PSNR of a held-out batch per mode, and how far the f16x2 / bf16
parameters drift from the f32-chain run).  The scene is the analytic shiny sphere of tests/golden/make_golden.py (re-stated
here), the optimiser torch.optim.Adam(fused=True) on the flat parameter blob."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import refnerf_pl_amd  # noqa: F401
from refnerf_pl_amd import configs, models, synthetic, train_utils, utils


def target(rays):
    o, d = rays["origins"].astype(np.float64), rays["viewdirs"].astype(np.float64)
    b = (o * d).sum(-1); c = (o * o).sum(-1) - 1.0; disc = b * b - c; hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0.0)); p = o + t[:, None] * d
    n = p / np.maximum(np.linalg.norm(p, axis=-1, keepdims=True), 1e-9)
    light = np.array([0.5, 0.6, 0.62]); light /= np.linalg.norm(light)
    refl = d - 2.0 * (d * n).sum(-1, keepdims=True) * n
    spec = np.maximum(0.0, refl @ light) ** 24 * 0.8
    col = (0.5 + 0.4 * n) * np.maximum(0.15, n @ light)[:, None] + spec[:, None]
    return np.where(hit[:, None], np.clip(col, 0.0, 1.0), 1.0).astype(np.float32)


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
N = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
batches = [synthetic.blender_rays(R, seed=9000 + i, center_frac=0.85) for i in range(steps)]
held = synthetic.blender_rays(4096, seed=123, center_frac=0.85)
out = {"steps": steps, "rays": R, "samples": N, "modes": {}}
final = {}
for mode in ("f32", "f16x2", "bf16"):
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")], [
        f"Model.num_prop_samples = {N}", f"Model.num_nerf_samples = {N}", f"Config.hip_train_precision = '{mode}'",
        f"Config.hip_bwd_precision = '{mode}'", "Config.hip_flat_grads = True", "Config.hip_fused_losses = True"])
    cfg = configs.Config()
    model = models.construct_model(None, cfg).to(dev)
    model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.0))
    opt = torch.optim.Adam([model.nerf_mlp.flat_parameter()], lr=5e-4, eps=1e-6, fused=True)
    curve = []
    t0 = time.time()
    for it, rd in enumerate(batches):
        model.train()
        rays = utils.rays_from_dict(rd, dev)
        batch = utils.Batch(rays=rays, rgb=target(rd))
        opt.zero_grad(set_to_none=True)
        rend, hist = model(rays, 1.0, False)
        total, terms, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
        total.backward()
        opt.step()
        model.nerf_mlp.mark_updated()
        if it % 25 == 0 or it == steps - 1:
            curve.append([it, float(terms["data"].detach())])
    torch.cuda.synchronize()
    el = time.time() - t0
    model.eval(); cfg.hip_precision = "f32"
    with torch.no_grad():
        r, _ = model(utils.rays_from_dict(held, dev), 1.0, False)
    mse = float(((r[1]["rgb"].cpu().numpy() - target(held)) ** 2).mean())
    final[mode] = model.nerf_mlp.flat_params().detach().cpu().numpy().copy()
    out["modes"][mode] = {"data_loss_curve": curve, "held_out_psnr_db": -10 * np.log10(mse), "seconds": el,
                          "finite": bool(np.isfinite(final[mode]).all())}
for mode in ("f16x2", "bf16"):
    d = final[mode] - final["f32"]
    out["modes"][mode]["param_rel_l2_vs_f32_chain_run"] = float(np.linalg.norm(d) / np.linalg.norm(final["f32"]))
print(json.dumps(out))
