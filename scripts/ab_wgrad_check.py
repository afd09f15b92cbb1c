"""A/B of library builds on the parameter gradients of one training step (golden training fixtures, the split-f16 chains, both
weight-gradient input modes): python scripts/ab_wgrad_check.py lib1.so lib2.so ... -- per parameter tensor, the rel-L2 distance of
every build's gradient from the FIRST build's."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/ab_wgrad_check"
FIXTURES = os.environ.get("REFNERF_AB_FIXTURES", "model_blender_sharp_train model_trained_train").split()
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    import refnerf_pl_amd  # noqa
    from refnerf_pl_amd import _hip, configs, models, train_utils, utils
    if sys.argv[2] != "-":
        _hip.LIB_PATH = os.path.join(ROOT, sys.argv[2])
    from helpers import load_golden, params_from_golden, rays_from_golden
    tag = os.path.basename(sys.argv[2]).replace(".so", "")
    os.makedirs(OUT, exist_ok=True)
    for name in FIXTURES:
        g = load_golden(name)
        bindings = [str(b) for b in g["bindings"] if str(b)]
        for wgrad in ("bf16x3", "f16"):
            configs.clear_config()
            configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")],
                                                    bindings + ["Config.hip_train_precision = 'f16x2'", "Config.hip_bwd_precision = 'f16x2'",
                                                                f"Config.hip_wgrad_mode = '{wgrad}'"])
            cfg = configs.Config()
            model = models.construct_model(utils.dummy_rays(), cfg).to("cuda:0").train()
            model.nerf_mlp.load_flat_params(params_from_golden(g))
            rays = utils.rays_from_dict(rays_from_golden(g), "cuda:0")
            batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
            rend, hist = model(rays, 1.0, False)
            total, terms, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
            total.backward()
            grads = torch.cat([p.grad.flatten() for p in model.nerf_mlp.ordered_parameters()]).cpu().numpy()
            np.save(os.path.join(OUT, f"{tag}.{name}.{wgrad}.npy"), grads)
            print(tag, name, wgrad, "rays", rays.origins.shape, "grad l2 %.9e" % float(np.linalg.norm(grads.astype(np.float64))), flush=True)
    # a full-width batch of the 2500-step weights (1024 rays x 128 samples: 4096 k-steps of the GEMM, real delta ranges)
    from helpers import trained_long_blob
    from refnerf_pl_amd import synthetic
    for wgrad in ("bf16x3", "f16"):
        configs.clear_config()
        configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")],
                                                ["Config.hip_train_precision = 'f16x2'", "Config.hip_bwd_precision = 'f16x2'", f"Config.hip_wgrad_mode = '{wgrad}'"])
        cfg = configs.Config()
        model = models.construct_model(utils.dummy_rays(), cfg).to("cuda:0").train()
        model.nerf_mlp.load_flat_params(trained_long_blob())
        rd = synthetic.blender_rays(1024, seed=3, center_frac=0.8)
        rays = utils.rays_from_dict(rd, "cuda:0")
        batch = utils.Batch(rays=rays, rgb=np.random.default_rng(5).random((1024, 3)).astype(np.float32))
        rend, hist = model(rays, 1.0, False)
        total, terms, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
        total.backward()
        grads = torch.cat([p.grad.flatten() for p in model.nerf_mlp.ordered_parameters()]).cpu().numpy()
        np.save(os.path.join(OUT, f"{tag}.big.{wgrad}.npy"), grads)
        print(tag, "big", wgrad, "grad l2 %.9e" % float(np.linalg.norm(grads.astype(np.float64))), flush=True)
else:
    import numpy as np
    sys.path.insert(0, ROOT)
    for lib in sys.argv[1:]:
        subprocess.call([sys.executable, __file__, "--child", lib])
    from refnerf_pl_amd import layout
    tags = [os.path.basename(l).replace(".so", "") for l in sys.argv[1:]]
    for name in FIXTURES + ["big"]:
        for wgrad in ("bf16x3", "f16"):
            a = np.load(os.path.join(OUT, f"{tags[0]}.{name}.{wgrad}.npy")).astype(np.float64)
            for t in tags[1:]:
                b = np.load(os.path.join(OUT, f"{t}.{name}.{wgrad}.npy")).astype(np.float64)
                print(f"== {name} {wgrad}: {t} vs {tags[0]}: rel-L2 {np.linalg.norm(a - b) / np.linalg.norm(a):.3e}")
                for s in layout.PARAM_SPECS:
                    w = slice(s.w_off, s.w_off + s.out_dim * s.in_dim); bb = slice(s.b_off, s.b_off + s.out_dim)
                    rw = np.linalg.norm(a[w] - b[w]) / max(np.linalg.norm(a[w]), 1e-30); rb = np.linalg.norm(a[bb] - b[bb]) / max(np.linalg.norm(a[bb]), 1e-30)
                    flag = "  <--" if max(rw, rb) > 1e-3 else ""
                    print(f"   {s.name:18s} W {rw:.2e}  b {rb:.2e}{flag}")
