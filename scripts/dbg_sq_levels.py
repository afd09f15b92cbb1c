"""GPU debug aid: the two levels of test_trained_long_split_chains_equal_f32_chains_from_the_same_step_function (white-noise
upstream gradients, identical step functions) in the f32 and the f16x2 chain mode through the C ABI, per-tensor gradient
differences.  Usage: python scripts/dbg_sq_levels.py [tag] [N0 N1]   (REFNERF_LEGACY_F16X2_TRAIN=1: the round-4 kernels).
DEBUG INFRASTRUCTURE: never imported by the product."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refnerf_pl_amd  # noqa: F401,E402
from refnerf_pl_amd import _hip as hip, layout  # noqa: E402
from helpers import load_golden, params_from_golden, rays_from_golden, cfg_from_bindings  # noqa: E402

DEV = torch.device("cuda:0")
F16X2 = hip.PREC_F16X2


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "trained_long"
    Ns = [int(a) for a in sys.argv[2:4]] if len(sys.argv) > 3 else [64, 96]
    g = load_golden(f"model_{tag}_train")
    P = torch.tensor(params_from_golden(g), device=DEV)
    rays = {k: torch.tensor(v, device=DEV) for k, v in rays_from_golden(g).items()}
    for k in ("radii", "near", "far"):
        rays[k] = rays[k].reshape(-1)
    R = rays["origins"].shape[0]
    packed = {prec: hip.pack_weights(P, precision=hip.level_image(prec, True)) for prec in (0, F16X2)}
    gen = torch.Generator().manual_seed(3)
    sd, w = torch.tensor([[0.0, 1.0]], device=DEV).repeat(R, 1), torch.ones((R, 1), device=DEV)
    for N in Ns:
        g_rgb = (torch.randn((R, 3), generator=gen) * 1e-2).to(DEV)
        g_w = (torch.randn((R, N), generator=gen) * 1e-3).to(DEV)
        grads, outs = {}, {}
        for prec in (0, F16X2):
            cfg = hip.default_cfg(n_samples=N, n_in=w.shape[1], training=1, compute_extras=0, **cfg_from_bindings(g["bindings"])[0])
            cfg.precision = prec
            if prec == F16X2 and os.environ.get("REFNERF_WGRAD_MODE") == "f16":
                cfg.wgrad_mode = hip.WGRAD_F16
            res = hip.level_forward(packed[prec], cfg, rays, sd, w, history=True, save_activations=True)
            out = torch.zeros(hip.NUM_PARAMS, device=DEV)
            hip.level_backward(packed[prec], cfg, rays, res, g_rgb, g_w, None, out)
            grads[prec], outs[prec] = out.cpu().numpy(), res
        a, b = grads[0], grads[F16X2]
        print(f"N = {N}, n_in = {w.shape[1]}, R = {R}: gradient rel diff between the chain modes {np.linalg.norm(a - b) / np.linalg.norm(a):.3e}  nonfinite {int((~np.isfinite(b)).sum())}")
        for s in layout.PARAM_SPECS:
            wa, wb = a[s.w_off:s.w_off + s.out_dim * s.in_dim], b[s.w_off:s.w_off + s.out_dim * s.in_dim]
            ba, bb = a[s.b_off:s.b_off + s.out_dim], b[s.b_off:s.b_off + s.out_dim]
            print(f"   {s.name:18s} |gW| {np.linalg.norm(wa):.3e} rel {np.linalg.norm(wa - wb) / max(np.linalg.norm(wa), 1e-30):.2e}   "
                  f"|gb| {np.linalg.norm(ba):.3e} rel {np.linalg.norm(ba - bb) / max(np.linalg.norm(ba), 1e-30):.2e}")
        for k in ("density", "rgb", "normals", "weights"):
            if k in outs[0] and outs[0][k] is not None:
                print(f"   fwd {k}: max diff {float((outs[0][k] - outs[F16X2][k]).abs().max()):.2e}")
        sd, w = outs[0]["sdist"].contiguous(), outs[0]["weights"].contiguous()


if __name__ == "__main__":
    main()
