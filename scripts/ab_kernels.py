"""Per-kernel times of the training step for several library builds (HIP events around each library call):
   python scripts/ab_kernels.py a.so b.so"""
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
    cfg0 = _hip.default_cfg(n_samples=N, n_in=1, training=1, compute_extras=0)
    res0 = _hip.level_forward(packed, cfg0, rays, sd, w, history=True, save_activations=True)
    cfg = _hip.default_cfg(n_samples=N, n_in=N, training=1, compute_extras=0)
    g_rgb = torch.randn((R, 3), device=dev) * 1e-3
    g_w = torch.randn((R, N), device=dev) * 1e-3
    g_np = torch.randn((R, N, 3), device=dev) * 1e-3
    grads = torch.zeros(_hip.NUM_PARAMS, device=dev)

    def timeit(fn, n=6):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    saved = {}

    def fwd():
        res = _hip.level_forward(packed, cfg, rays, res0["sdist"], res0["weights"], history=True, save_activations=True)
        saved.update({k: res[k] for k in ("sdist", "density", "rgb", "weights", "activations")})
    t_f = timeit(fwd)

    def bwd():
        _hip.level_backward(packed, cfg, rays, saved, g_rgb, g_w, g_np, grads)
    t_b = timeit(bwd)
    print(f"{lib}: level-1 training forward {t_f:.2f} ms, backward + wgrad {t_b:.2f} ms", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for rep in range(2):
            for lib in sys.argv[1:]:
                subprocess.call([sys.executable, __file__, "--child", lib])
