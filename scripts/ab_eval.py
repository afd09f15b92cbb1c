"""A/B of the eval forward (f32 and bf16 modes) with several library builds: python scripts/ab_eval.py a.so b.so"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(lib):
    import torch
    import refnerf_pl_amd  # noqa: F401
    from refnerf_pl_amd import _hip
    _hip.LIB_PATH = os.path.join(ROOT, lib)
    from refnerf_pl_amd import configs, models, synthetic, utils
    dev = torch.device("cuda", 0)
    for prec in ("f32", "bf16"):
        configs.clear_config()
        configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")],
                                                [f"Config.hip_precision = '{prec}'"])
        cfg = configs.Config()
        model = models.construct_model(None, cfg).to(dev).eval()
        model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
        rays = utils.rays_from_dict(synthetic.blender_rays(4096, seed=1, center_frac=0.5), dev)
        n = 20 if prec == "f32" else 200
        with torch.no_grad():
            for _ in range(n // 4):
                out = model(rays, 1.0, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                out = model(rays, 1.0, True)
            torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        print(f"{lib} {prec} eval {ms:.3f} ms/step  rgb sum {float(out[0][-1]['rgb'].double().sum()):.9f}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for rep in range(2):
            for lib in sys.argv[1:]:
                subprocess.call([sys.executable, __file__, "--child", lib])
