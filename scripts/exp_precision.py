"""Accuracy of the throughput (bf16 / f16) eval mode against the f32 parity mode of the same build, on the trained-like
weights (tests/golden/trained_blob.npz) and on the random-init bench weights, for several library builds:
python scripts/exp_precision.py a.so b.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(lib):
    import numpy as np
    import torch
    import refnerf_pl_amd  # noqa: F401
    from refnerf_pl_amd import _hip
    _hip.LIB_PATH = os.path.join(ROOT, lib)
    from refnerf_pl_amd import configs, models, synthetic, utils
    from helpers import trained_blob
    dev = torch.device("cuda", 0)
    for wname, blob, rk in (("trained", trained_blob(), dict(seed=3, center_frac=0.8)),
                            ("random-init", synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0), dict(seed=1, center_frac=0.5))):
        outs = {}
        for prec in ("f32", "bf16", "f16"):
            configs.clear_config()
            configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")],
                                                    [f"Config.hip_precision = '{prec}'"])
            cfg = configs.Config()
            model = models.construct_model(None, cfg).to(dev).eval()
            model.nerf_mlp.load_flat_params(blob)
            rays = utils.rays_from_dict(synthetic.blender_rays(2048, **rk), dev)
            with torch.no_grad():
                outs[prec] = model(rays, 1.0, True)
            torch.cuda.synchronize()
        for mode, L in ((m, l) for m in ("bf16", "f16") for l in range(2)):
            a, b = outs["f32"][0][L]["rgb"].double(), outs[mode][0][L]["rgb"].double()
            d = (a - b).abs()
            ps = (outs["f32"][1][L]["rgb"] - outs[mode][1][L]["rgb"]).abs().max()
            dn = (outs["f32"][1][L]["density"] - outs[mode][1][L]["density"]).abs().max()
            same = (outs["f32"][1][L]["sdist"] == outs[mode][1][L]["sdist"]).double().mean()
            print(f"{lib:22s} {mode:4s} {wname:11s} L{L}: rgb L-inf {float(d.max()):.2e}  mean {float(d.mean()):.2e}  99.9% {float(d.flatten().quantile(0.999)):.2e}  "
                  f"PSNR {float(-10 * torch.log10((d ** 2).mean())):.1f} dB  per-sample rgb {float(ps):.2e} density {float(dn):.2e}  sdist== {float(same):.4f}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for lib in sys.argv[1:]:
            subprocess.call([sys.executable, __file__, "--child", lib])
