import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refnerf_pl_amd
from refnerf_pl_amd import _hip, synthetic
from test_hip_parity import run_hip_model
P = synthetic.make_params(0, 0.05, 20.0)
rays = synthetic.blender_rays(3, seed=1, center_frac=0.3)
a = run_hip_model(_hip, P, rays, {}, dict(num_levels=1, num_nerf_samples=128), precision=0)
b = run_hip_model(_hip, P, rays, {}, dict(num_levels=1, num_nerf_samples=128), precision=1)
for k in ("density","roughness","tint","grad_pred","normals_pred","diffuse","specular","rgb","weights"):
    x=b[0][k]; y=a[0][k]
    print(k, "nan frac", np.isnan(x).mean(), "first ray nan idx", np.where(np.isnan(x[0].reshape(128,-1)).any(-1))[0][:20], "maxdiff(finite)", np.nanmax(np.abs(x-y)))
