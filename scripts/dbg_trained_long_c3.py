import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import trained_long_blob
import refnerf_pl_amd
from refnerf_pl_amd import synthetic, _hip as hip
from test_hip_parity import run_hip_model
from oracle import oracle as O
P = trained_long_blob()
for N in (128, 160, 192, 256):
    rays = synthetic.blender_rays(8192, seed=3, center_frac=0.8)
    lv = dict(num_prop_samples=N, num_nerf_samples=N)
    sub = {k: v[:256] for k, v in rays.items()}
    ref = O.model_forward(P, sub, **lv)
    for tag, rr in (("8192 rays", rays), ("256 rays (plain kernel)", sub)):
        x = run_hip_model(hip, P, rr, {}, lv, precision=3)
        y = run_hip_model(hip, P, rr, {}, lv, precision=0)
        e = np.abs(x[1]["r_rgb"][:256] - ref[1]["r_rgb"]); e32 = np.abs(y[1]["r_rgb"][:256] - ref[1]["r_rgb"])
        i = int(e.max(-1).argmax())
        print(f"N={N} {tag}: f16x2 {e.max():.2e} (ray {i}: acc {ref[1]['r_acc'][i]:.4f}, f32-mode err on it {e32[i].max():.2e}), f32 {e32.max():.2e}, "
              f"density max {x[1]['density'][:256].max():.1f}, weights diff {np.abs(x[1]['weights'][:256]-ref[1]['weights']).max():.2e} f32 {np.abs(y[1]['weights'][:256]-ref[1]['weights']).max():.2e}")
