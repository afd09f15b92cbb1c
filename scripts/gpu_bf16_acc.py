import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refnerf_pl_amd
from refnerf_pl_amd import _hip, synthetic
from test_hip_parity import run_hip_model
for nm, P, rays in (("blender sharp", synthetic.make_params(0, 0.05, 20.0), synthetic.blender_rays(4096, seed=1, center_frac=0.5)),
                    ("blender init", synthetic.make_params(0, 0.0, 1.0), synthetic.blender_rays(4096, seed=2, center_frac=0.5)),
                    ("blender shiny", synthetic.make_params(3, 0.05, 20.0, -6.0), synthetic.blender_rays(4096, seed=3, center_frac=0.5)),
                    ("llff sharp", synthetic.make_params(4, 0.05, 20.0), synthetic.llff_rays(4096, seed=5))):
    a = run_hip_model(_hip, P, rays, {}, {}, precision=0)
    b = run_hip_model(_hip, P, rays, {}, {}, precision=1)
    for L in range(2):
        d = np.abs(a[L]["r_rgb"] - b[L]["r_rgb"])
        mse = float(np.mean((a[L]["r_rgb"] - b[L]["r_rgb"]) ** 2))
        print(nm, "L", L, "rgb Linf %.2e mean %.2e psnr %.1f dB idx_eq %.4f acc range %.2f..%.2f" % (d.max(), d.mean(), -10 * np.log10(mse + 1e-30), np.mean(a[L]["bin_idx"] == b[L]["bin_idx"]), a[L]["r_acc"].min(), a[L]["r_acc"].max()))
