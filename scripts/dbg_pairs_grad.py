"""per-tensor gradient differences between the chain modes on a golden training fixture (GPU):
  python scripts/dbg_pairs_grad.py [model_trained_long_train]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import refnerf_pl_amd
from refnerf_pl_amd import configs, layout, models, train_utils, utils
from helpers import load_golden, params_from_golden, rays_from_golden
name = sys.argv[1] if len(sys.argv) > 1 else "model_trained_long_train"
g = load_golden(name)
DEV = "cuda:0"
res = {}
for chains in ("f32", "f16x2"):
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")],
                                            [str(b) for b in g["bindings"] if str(b)] + [f"Config.hip_train_precision = '{chains}'", f"Config.hip_bwd_precision = '{chains}'"])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).train()
    model.nerf_mlp.load_flat_params(params_from_golden(g))
    rays = utils.rays_from_dict(rays_from_golden(g), DEV)
    batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
    rend, hist = model(rays, 1.0, False)
    total, terms, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
    total.backward()
    res[chains] = torch.cat([p.grad.flatten() for p in model.nerf_mlp.ordered_parameters()]).cpu().numpy()
a, b = res["f32"], res["f16x2"]
print("whole: rel", np.linalg.norm(a - b) / np.linalg.norm(a), "vs reference (sub):", np.linalg.norm(b[::97] - g["grads_sub"]) / np.linalg.norm(g["grads_sub"]),
      "f32 vs reference:", np.linalg.norm(a[::97] - g["grads_sub"]) / np.linalg.norm(g["grads_sub"]))
rows = []
for s in layout.PARAM_SPECS:
    for tag, off, n in (("w", s.w_off, s.out_dim * s.in_dim), ("b", s.b_off, s.out_dim)):
        x, y = a[off:off + n], b[off:off + n]
        rows.append((np.linalg.norm(x - y), s.name + "." + tag, np.linalg.norm(x), np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-30), bool(np.isfinite(y).all())))
for r in sorted(rows, reverse=True)[:14]:
    print("  %-28s |diff| %.3e  |g| %.3e  rel %.2e finite %s" % (r[1], r[0], r[2], r[3], r[4]))
