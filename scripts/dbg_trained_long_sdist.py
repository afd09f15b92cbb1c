"""f32 vs split-f16 training forward on the long-trained fixture: do the level-1 sample positions differ?"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from helpers import load_golden, params_from_golden, rays_from_golden
import refnerf_pl_amd
from refnerf_pl_amd import configs, models, utils
g = load_golden("model_trained_long_train")
res = {}
for fwd in ("f32", "f16x2"):
    configs.clear_config()
    configs.parse_config_files_and_bindings(["configs/refnerf_blender.gin"], [str(b) for b in g["bindings"] if str(b)] + [f"Config.hip_train_precision = '{fwd}'", f"Config.hip_bwd_precision = '{fwd}'"])
    model = models.construct_model(utils.dummy_rays(), configs.Config()).to("cuda:0").train()
    model.nerf_mlp.load_flat_params(params_from_golden(g))
    rays = utils.rays_from_dict(rays_from_golden(g), "cuda:0")
    rend, hist = model(rays, 1.0, False)
    res[fwd] = ([h["sdist"].detach().cpu().numpy() for h in hist], [h["weights"].detach().cpu().numpy() for h in hist], [np.asarray(b.cpu()) for b in model.last_bin_idx])
for L in range(2):
    ds = np.abs(res["f32"][0][L] - res["f16x2"][0][L]); dw = np.abs(res["f32"][1][L] - res["f16x2"][1][L])
    same_bin = np.mean(res["f32"][2][L] == res["f16x2"][2][L])
    print(f"level {L}: sdist max diff {ds.max():.2e} (# > 1e-6: {(ds > 1e-6).sum()}), weights max diff {dw.max():.2e}, same bin idx {same_bin:.6f}")
    r = g[f"L{L}_h_sdist"]
    for k in ("f32", "f16x2"):
        d = np.abs(res[k][0][L] - r.reshape(res[k][0][L].shape))
        print(f"    {k} vs reference sdist: max {d.max():.2e}, # > 1e-6: {(d > 1e-6).sum()}")
