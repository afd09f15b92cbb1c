"""A/B of backward + weight-gradient time (and gradient agreement) for several library builds:
   python scripts/ab_bwd.py a.so b.so"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(lib):
    import torch
    import refnerf_pl_amd  # noqa: F401
    from refnerf_pl_amd import _hip
    _hip.LIB_PATH = os.path.join(ROOT, lib)
    from refnerf_pl_amd import synthetic
    dev = "cuda:0"
    R, N = 4096, 128
    P = torch.tensor(synthetic.make_params(0, 0.05, 20.0), device=dev)
    rays = {k: torch.tensor(v, device=dev) for k, v in synthetic.blender_rays(R, seed=1, center_frac=0.5).items()}
    for k in ("radii", "near", "far"):
        rays[k] = rays[k].reshape(-1)
    packed = _hip.pack_weights(P, precision=0)
    sd = torch.tensor([[0.0, 1.0]], device=dev).repeat(R, 1)
    w = torch.ones((R, 1), device=dev)
    cfg = _hip.default_cfg(n_samples=N, n_in=1, training=1, compute_extras=0)
    g = torch.Generator().manual_seed(0)
    g_rgb = (torch.randn((R, 3), generator=g) * 1e-3).to(dev)
    g_w = (torch.randn((R, N), generator=g) * 1e-3).to(dev)
    g_np = (torch.randn((R, N, 3), generator=g) * 1e-3).to(dev)
    res = _hip.level_forward(packed, cfg, rays, sd, w, history=True, save_activations=True)
    cfg.precision = int(os.environ.get("AB_BWD_PREC", "0"))       # 1: bf16 chains in the backward
    grads = torch.zeros(_hip.NUM_PARAMS, device=dev)
    _hip.level_backward(packed, cfg, rays, res, g_rgb, g_w, g_np, grads)
    ref = grads.double().cpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    for _ in range(n):
        grads.zero_()
        _hip.level_backward(packed, cfg, rays, res, g_rgb, g_w, g_np, grads)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(f"{lib}: backward + wgrad {ms:.2f} ms  |g| {float(ref.norm()):.9e}  checksum {float(ref.sum()):.9e}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for rep in range(2):
            for lib in sys.argv[1:]:
                subprocess.call([sys.executable, __file__, "--child", lib])
