import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import refnerf_pl_amd
from refnerf_pl_amd import _hip, layout, synthetic
dev = "cuda:0"
rays = {k: torch.tensor(v, device=dev) for k, v in synthetic.blender_rays(64, seed=2, center_frac=0.6).items()}
for k in ("radii", "near", "far"): rays[k] = rays[k].reshape(-1)
R, N = 64, 64
sd = torch.tensor([[0.0, 1.0]], device=dev).repeat(R, 1); w = torch.ones((R, 1), device=dev)
S = R * N; UNITS = 4525
def row(a, u):
    s = np.arange(S)
    return a[(s >> 6) * UNITS * 64 + u * 64 + (s & 63)]
for scale in (30.0, 300.0, 3000.0, 30000.0):
    P = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=1.0)
    for name in ("spatial_net.2", "spatial_net.3"):
        sp = layout.SPEC_BY_NAME[name]; P[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim] *= scale
    nxt = layout.SPEC_BY_NAME["spatial_net.4"]; P[nxt.w_off:nxt.w_off + nxt.out_dim * nxt.in_dim] /= scale * scale
    packed = _hip.pack_weights(torch.tensor(P, device=dev), precision=0)
    out = {}
    for prec in (0, 3):
        cfg = _hip.default_cfg(n_samples=N, n_in=1, training=1, compute_extras=0); cfg.precision = prec
        res = _hip.level_forward(packed, cfg, rays, sd, w, history=True, save_activations=True)
        a = res["activations"].view(torch.float32).cpu().numpy()
        x3 = np.stack([row(a, 96 + 3 * 256 + j) for j in range(256)])     # input of layer 4 = output of layer 3
        out[prec] = (x3, res["r_rgb"].cpu().numpy())
    print(f"scale {scale:g}: max x3 (f32) {np.abs(out[0][0]).max():.4g}; f16x2 x3 max {np.nanmax(np.abs(out[3][0])):.4g} finite {np.isfinite(out[3][0]).all()}; "
          f"x3 max rel diff {np.nanmax(np.abs(out[3][0] - out[0][0])) / np.abs(out[0][0]).max():.2e}; rgb diff {np.nanmax(np.abs(out[3][1] - out[0][1])):.2e} finite {np.isfinite(out[3][1]).all()}")
