"""Debug aid: per-tensor relative error of the HIP backward vs the oracle on a golden training fixture."""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import refnerf_pl_amd
from refnerf_pl_amd import _hip as hip, layout
from oracle import oracle as O
from helpers import cfg_from_bindings, load_golden, params_from_golden, rays_from_golden
import test_hip_parity as T
for name in T.TRAIN_CASES:
    g = load_golden(name); P = params_from_golden(g); rays = rays_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    mults = ((0.1, 1.0), (0.01, 0.1), (3e-5, 3e-4))
    losses, grads = T._hip_train_step(hip, P, rays, g["gt_rgb"], rays["lossmult"], kw, lv, mults)
    ol, og, _ = O.model_train(P, rays, g["gt_rgb"], **lv, **kw)
    print(name, losses, ol)
    print(" total rel", np.linalg.norm(grads - og) / np.linalg.norm(og), "vs golden", np.linalg.norm(grads[::97] - g["grads_sub"]) / np.linalg.norm(g["grads_sub"]),
          "oracle vs golden", np.linalg.norm(og[::97] - g["grads_sub"]) / np.linalg.norm(g["grads_sub"]))
    for s in layout.PARAM_SPECS:
        nw = s.out_dim * s.in_dim
        a, b = grads[s.w_off:s.w_off + nw], og[s.w_off:s.w_off + nw]
        c, d = grads[s.b_off:s.b_off + s.out_dim], og[s.b_off:s.b_off + s.out_dim]
        print(f"  {s.name:28s} W {np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30):.2e} (|g|={np.linalg.norm(b):.2e})  b {np.linalg.norm(c-d)/max(np.linalg.norm(d),1e-30):.2e}")
