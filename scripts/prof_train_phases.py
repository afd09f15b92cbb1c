"""REFNERF_PROF=1 cycle stamps of the training forward + backward of one level (C2 shape): python scripts/prof_train_phases.py <precision 0|1|3>
forward slots: 1 resample | 2 pass start | 3 IPE | 4 spatial trunk | 5 heads | 6 density normals | 7 IDE | 8 dir trunk | 9 rgb | 10 colour
backward slots (see refnerf_hip.hip): prologue | heads recompute | rgb + colour head | seed | dir chain | IDE / heads | spatial chain"""
import os, sys
os.environ["REFNERF_PROF"] = "1"
sys.path.insert(0, os.getcwd())
import torch
import refnerf_pl_amd
from refnerf_pl_amd import _hip, synthetic
if os.environ.get("REFNERF_LIB"):
    _hip.LIB_PATH = os.path.join(os.getcwd(), os.environ["REFNERF_LIB"])
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda:0"; R, N = 4096, 128
P = torch.tensor(synthetic.make_params(0, 0.05, 20.0), device=dev)
rays = {k: torch.tensor(v, device=dev) for k, v in synthetic.blender_rays(R, seed=1, center_frac=0.5).items()}
for k in ("radii", "near", "far"): rays[k] = rays[k].reshape(-1)
packed = _hip.pack_weights(P, precision=_hip.level_image(prec, True))
sd = torch.tensor([[0.0, 1.0]], device=dev).repeat(R, 1); w = torch.ones((R, 1), device=dev)
g_rgb = torch.randn((R, 3), device=dev) * 1e-3; g_w = torch.randn((R, N), device=dev) * 1e-3; g_np = torch.randn((R, N, 3), device=dev) * 1e-3
cfg = _hip.default_cfg(n_samples=N, n_in=1, training=1, compute_extras=0)
cfg.precision = prec
for it in range(2):
    print("== forward, precision", prec, file=sys.stderr)
    res = _hip.level_forward(packed, cfg, rays, sd, w, history=True, save_activations=True)
    grads = torch.zeros(_hip.NUM_PARAMS, device=dev)
    print("== backward", file=sys.stderr)
    _hip.level_backward(packed, cfg, rays, res, g_rgb, g_w, g_np, grads)
    torch.cuda.synchronize()
