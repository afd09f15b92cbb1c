"""Ad-hoc GPU check: HIP level kernel vs oracle on a golden case (prints diffs)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refnerf_pl_amd  # noqa
from refnerf_pl_amd import _hip, synthetic
from oracle import oracle as O
from helpers import cfg_from_bindings, load_golden, params_from_golden, rays_from_golden

dev = torch.device("cuda:0")
names = sys.argv[1:] or ["model_blender_sharp_eval", "model_c1_eval", "model_llff_linear_eval"]
for name in names:
    g = load_golden(name)
    P = params_from_golden(g)
    rays = rays_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    oracle_out = O.model_forward(P, rays, **lv, **kw)
    packed = _hip.pack_weights(torch.tensor(P, device=dev))
    torch.cuda.synchronize()
    drays = {k: torch.tensor(v, device=dev).reshape(v.shape[0], -1).squeeze(-1) if k in ("radii", "near", "far") else torch.tensor(v, device=dev) for k, v in rays.items()}
    R = rays["origins"].shape[0]
    sdist = torch.tensor([[0.0, 1.0]], device=dev).repeat(R, 1)
    weights = torch.ones((R, 1), device=dev)
    nl = lv.get("num_levels", 2)
    for L in range(nl):
        n = lv.get("num_prop_samples", 128) if L < nl - 1 else lv.get("num_nerf_samples", 128)
        cfg = _hip.default_cfg(n_samples=n, n_in=weights.shape[1], **kw)
        t0 = time.time()
        res = _hip.level_forward(packed, cfg, drays, sdist, weights)
        torch.cuda.synchronize()
        dt = time.time() - t0
        ref = oracle_out[L]
        line = []
        for k in ("sdist", "weights", "density", "rgb", "normals_pred", "grad_pred", "roughness", "diffuse", "specular", "tint",
                  "r_rgb", "r_diffuse", "r_specular", "r_distance", "r_acc", "r_normals_pred", "r_tint", "r_roughness", "r_distance_mean", "r_percentiles"):
            a = res[k].cpu().numpy()
            line.append(f"{k}:{np.abs(a - ref[k].reshape(a.shape)).max():.1e}")
        idx_eq = np.mean(res["bin_idx"].cpu().numpy() == ref["bin_idx"])
        print(name, "L", L, f"{dt*1e3:.2f} ms", "idx_eq", idx_eq, " ".join(line))
        gold = g[f"L{L}_r_rgb"]
        print("   vs reference golden rgb Linf:", np.abs(res["r_rgb"].cpu().numpy() - gold).max())
        sdist, weights = res["sdist"], res["weights"]
