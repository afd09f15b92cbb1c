"""A/B timing of the training step with two builds of the library:
   python scripts/ab_train.py ab/libA.so ab/libB.so [rays samples]
Each library runs in its own child process (the library is loaded once per process)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(lib, R, N):
    import torch
    import refnerf_pl_amd  # noqa: F401
    from refnerf_pl_amd import _hip
    _hip.LIB_PATH = os.path.join(ROOT, lib)
    from refnerf_pl_amd import configs, models, synthetic, train_utils, utils
    dev = torch.device("cuda", 0)
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")], [
        f"Model.num_prop_samples = {N}", f"Model.num_nerf_samples = {N}", f"Config.batch_size = {R}",
        "Config.hip_precision = '%s'" % os.environ.get("AB_PREC", "f32"),
        "Config.hip_train_precision = '%s'" % os.environ.get("AB_TRAIN_PREC", "f32"),
        "Config.hip_bwd_precision = '%s'" % os.environ.get("AB_TRAIN_PREC", "f32")])
    cfg = configs.Config()
    model = models.construct_model(None, cfg).to(dev)
    model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
    rays = utils.rays_from_dict(synthetic.blender_rays(R, seed=1, center_frac=0.5), dev)
    batch = utils.Batch(rays=rays, rgb=synthetic.target_rgb(R, seed=7))
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    model.train()

    def step():
        opt.zero_grad(set_to_none=True)
        renderings, history = model(rays, 1.0, False)
        total, _, _ = train_utils.compute_losses(model, batch, rays, renderings, history, cfg)
        total.backward()
        opt.step()
        return total
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    g = torch.cat([p.grad.flatten() for p in model.parameters()])
    print(f"{lib} {R}x{N} train {ms:.2f} ms/step  {R*N*2/ms*1e3:.3e} rs/s  loss {float(loss):.6f} |g| {float(g.norm()):.6e}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        libs = [a for a in sys.argv[1:] if a.endswith(".so")]
        rest = [a for a in sys.argv[1:] if not a.endswith(".so")] or ["4096", "128"]
        for rep in range(2):
            for lib in libs:
                subprocess.call([sys.executable, __file__, "--child", lib] + rest)
