"""bit-level A/B of library builds on one training level (forward + backward, split-f16 chains): prints checksums that must
agree between builds whose arithmetic is meant to be identical.  python scripts/ab_train_check.py lib1.so lib2.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    import refnerf_pl_amd  # noqa
    from refnerf_pl_amd import _hip, synthetic
    if sys.argv[2] != "-":
        _hip.LIB_PATH = os.path.join(ROOT, sys.argv[2])
    from helpers import trained_long_blob
    dev = "cuda:0"; R, N = 1024, 128
    P = torch.tensor(trained_long_blob(), device=dev)
    rays = {k: torch.tensor(v, device=dev) for k, v in synthetic.blender_rays(R, seed=3, center_frac=0.8).items()}
    for k in ("radii", "near", "far"): rays[k] = rays[k].reshape(-1)
    packed = _hip.pack_weights(P, precision=0)
    sd = torch.tensor([[0.0, 1.0]], device=dev).repeat(R, 1); w = torch.ones((R, 1), device=dev)
    gen = torch.Generator().manual_seed(1)
    for lvl in range(2):
        g_rgb = (torch.randn((R, 3), generator=gen) * 1e-3).to(dev); g_w = (torch.randn((R, N), generator=gen) * 1e-4).to(dev)
        g_np = (torch.randn((R, N, 3), generator=gen) * 1e-4).to(dev)
        cfg = _hip.default_cfg(n_samples=N, n_in=w.shape[1], training=1, compute_extras=1, precision=3)
        res = _hip.level_forward(packed, cfg, rays, sd, w, history=True, save_activations=True)
        grads = torch.zeros(_hip.NUM_PARAMS, device=dev)
        _hip.level_backward(packed, cfg, rays, res, g_rgb, g_w, g_np, grads)
        torch.cuda.synchronize()
        print("L%d rgb %.12f normals %.12f density %.10f weights %.12f | grad sum %.12e l2 %.12e" % (
            lvl, float(res["r_rgb"].double().sum()), float(res["normals"].double().abs().sum()), float(res["density"].double().sum()),
            float(res["weights"].double().sum()), float(grads.double().sum()), float(grads.double().norm())), flush=True)
        sd, w = res["sdist"].contiguous(), res["weights"].contiguous()
else:
    for lib in sys.argv[1:]:
        print("==", lib, flush=True)
        subprocess.call([sys.executable, __file__, "--child", lib])
