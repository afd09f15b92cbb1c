import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import refnerf_pl_amd
from refnerf_pl_amd import configs, layout, models, train_utils, utils
from helpers import load_golden, params_from_golden, rays_from_golden
name = "model_blender_sharp_train"
g = load_golden(name)
bindings = [str(b) for b in g["bindings"] if str(b)]
res = {}
for bwd in ("f32", "f16x2"):
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")], bindings + ["Config.hip_train_precision = 'f32'", f"Config.hip_bwd_precision = '{bwd}'"])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to("cuda:0").train()
    model.nerf_mlp.load_flat_params(params_from_golden(g))
    rays = utils.rays_from_dict(rays_from_golden(g), "cuda:0")
    batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
    rend, hist = model(rays, 1.0, False)
    total, terms, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
    total.backward()
    res[bwd] = torch.cat([p.grad.flatten() for p in model.nerf_mlp.ordered_parameters()]).cpu().numpy()
a, b = res["f32"], res["f16x2"]
for s in layout.PARAM_SPECS:
    for nm, off, n in (("w", s.w_off, s.out_dim * s.in_dim), ("b", s.b_off, s.out_dim)):
        x, y = a[off:off + n], b[off:off + n]
        rel = np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-30)
        if rel > 2e-4:
            extra = ""
            if nm == "w" and s.in_dim > 256:
                X, Y = x.reshape(s.out_dim, s.in_dim), y.reshape(s.out_dim, s.in_dim)
                extra = " cols<256 rel %.1e, cols>=256 rel %.1e" % (np.linalg.norm(X[:, :256] - Y[:, :256]) / np.linalg.norm(X[:, :256]), np.linalg.norm(X[:, 256:] - Y[:, 256:]) / np.linalg.norm(X[:, 256:]))
            print("%-22s %s rel %.2e |x| %.3e%s" % (s.name, nm, rel, np.linalg.norm(x), extra))
print("total rel", np.linalg.norm(a - b) / np.linalg.norm(a))
