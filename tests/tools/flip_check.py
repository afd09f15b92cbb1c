"""Are the 1e-4-level gradient differences vs the oracle ReLU flips?  Perturb the weights by 1 ulp-ish noise:
a smooth function changes its gradient by ~1e-7; isolated flips show up as ~1e-4 jumps, as they do vs the oracle."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")  # run from the repo root
import refnerf_pl_amd
from refnerf_pl_amd import _hip as hip, synthetic
import test_hip_parity as T
from oracle import oracle as O
R = 32
P = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0)
rays = {k: v[:R] for k, v in synthetic.blender_rays(64, seed=1, center_frac=0.4).items()}
gt = synthetic.target_rgb(R, seed=5)
lm = np.ones((R, 1), np.float32); rays["lossmult"] = lm
lv = dict(num_levels=2, num_prop_samples=64, num_nerf_samples=64)
mults = ((0.1, 1.0), (0.01, 0.1), (3e-5, 3e-4))
_, g0 = T._hip_train_step(hip, P, rays, gt, lm, {}, lv, mults)
_, og = O.model_train(P, rays, gt, **lv)[:2]
print("hip vs oracle", np.linalg.norm(g0 - og) / np.linalg.norm(og))
rng = np.random.default_rng(0)
for t in range(4):
    P2 = (P * (1.0 + 1.2e-7 * rng.standard_normal(P.shape))).astype(np.float32)
    _, g1 = T._hip_train_step(hip, P2, rays, gt, lm, {}, lv, mults)
    _, og1 = O.model_train(P2, rays, gt, **lv)[:2]
    print(f"perturbed {t}: hip vs hip0 {np.linalg.norm(g1 - g0) / np.linalg.norm(g0):.2e}   oracle vs oracle0 {np.linalg.norm(og1 - og) / np.linalg.norm(og):.2e}   hip vs oracle {np.linalg.norm(g1 - og1) / np.linalg.norm(og1):.2e}")
