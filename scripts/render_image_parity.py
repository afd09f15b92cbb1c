"""A whole 800 x 800 view through models.render_image (157 chunks of 4096 rays) in the f16x2 and the f32 mode on the 2500-step
trained-like weights: python scripts/render_image_parity.py -> one JSON line (max |diff|, PSNR between the modes, times)."""
import json, os, sys, time, functools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import refnerf_pl_amd  # noqa: F401
from refnerf_pl_amd import camera_utils, configs, models, synthetic, utils
dev = torch.device("cuda:0")
configs.clear_config()
configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")], [])
cfg = configs.Config()
model = models.construct_model(None, cfg).to(dev).eval()
model.nerf_mlp.load_flat_params(np.load(os.path.join(ROOT, "tests", "golden", "trained_long_blob.npz"))["blob_f32"])
rd = synthetic.blender_image_rays(800, 800) if hasattr(synthetic, "blender_image_rays") else None
if rd is None:
    H = W = 800
    rd = synthetic.blender_rays(H * W, seed=77, center_frac=0.75)
    rays = utils.rays_from_dict({k: v.reshape(H, W, -1) for k, v in rd.items()}, dev)
else:
    rays = utils.rays_from_dict(rd, dev)
out = {}
imgs = {}
for prec in ("f32", "f16x2"):
    cfg.hip_precision = prec
    fn = functools.partial(model, train_frac=1.0, compute_extras=True)
    with torch.no_grad():
        models.render_image(fn, rays, cfg)          # warm-up (weight image, allocator)
        torch.cuda.synchronize(); t0 = time.time()
        r = models.render_image(fn, rays, cfg)
        torch.cuda.synchronize()
    out[prec + "_seconds"] = time.time() - t0
    imgs[prec] = {k: r[k].float().cpu().numpy() for k in ("rgb", "acc", "distance_mean") if k in r}
for k in imgs["f32"]:
    d = imgs["f16x2"][k].astype(np.float64) - imgs["f32"][k]
    out[k + "_max_abs_diff"] = float(np.abs(d).max())
    if k == "rgb":
        out["rgb_psnr_between_modes_db"] = float(-10 * np.log10(max(np.mean(d ** 2), 1e-30)))
        out["rgb_pixels_over_1e-4"] = int((np.abs(d).max(-1) > 1e-4).sum())
out["pixels"] = int(np.prod(imgs["f32"]["rgb"].shape[:-1]))
print(json.dumps(out))
