"""bf16 level kernel: accuracy vs the fp32 HIP path / oracle and timing."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refnerf_pl_amd
from refnerf_pl_amd import _hip, synthetic
from helpers import load_golden, params_from_golden, rays_from_golden, cfg_from_bindings
from test_hip_parity import run_hip_model
dev = "cuda:0"
for name in ["model_blender_sharp_eval", "model_c1_eval", "model_llff_linear_eval"]:
    g = load_golden(name); P = params_from_golden(g); rays = rays_from_golden(g); kw, lv = cfg_from_bindings(g["bindings"])
    a = run_hip_model(_hip, P, rays, kw, lv, precision=0)
    b = run_hip_model(_hip, P, rays, kw, lv, precision=1)
    for L in range(len(a)):
        line = [f"{k}:{np.abs(a[L][k]-b[L][k]).max():.1e}" for k in ("sdist","weights","density","rgb","normals_pred","roughness","r_rgb","r_acc","r_distance")]
        print(name, "L", L, "idx_eq", np.mean(a[L]["bin_idx"]==b[L]["bin_idx"]), " ".join(line), "golden rgb", np.abs(b[L]["r_rgb"]-g[f"L{L}_r_rgb"]).max())
# timing at C2
P = synthetic.make_params(0, 0.05, 20.0)
rays = synthetic.blender_rays(4096, seed=1, center_frac=0.5)
for prec in (0, 1):
    packed = _hip.pack_weights(torch.tensor(P, device=dev), precision=prec)
    from test_hip_parity import dev_rays
    r = dev_rays(rays)
    sd = torch.tensor([[0.0, 1.0]], device=dev).repeat(4096, 1); w = torch.ones((4096, 1), device=dev)
    cfg = _hip.default_cfg(n_samples=128, n_in=1, precision=prec)
    for _ in range(2): res = _hip.level_forward(packed, cfg, r, sd, w)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): res = _hip.level_forward(packed, cfg, r, sd, w)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print("prec", prec, f"{dt*1e3:.3f} ms/level  {4096*128/dt:.3e} samples/s  {4096*128*2211840/dt/1e12:.1f} TFLOP/s")
