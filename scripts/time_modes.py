"""Eval-forward timing of the arithmetic modes at a BASELINE shape: python scripts/time_modes.py [rays] [samples] [modes...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import refnerf_pl_amd  # noqa: F401
from refnerf_pl_amd import _hip, configs, models, synthetic, utils

if os.environ.get("REFNERF_LIB"):              # an A/B build (scripts/build_main_variant.sh) instead of the in-tree library
    _hip.LIB_PATH = os.path.join(ROOT, os.environ["REFNERF_LIB"])
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
modes = sys.argv[3:] or ["f16x2", "f16", "bf16", "f32"]
dev = torch.device("cuda", 0)
configs.clear_config()
configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")],
                                        [f"Model.num_prop_samples = {N}", f"Model.num_nerf_samples = {N}"])
cfg = configs.Config()
model = models.construct_model(None, cfg).to(dev).eval()
_blob = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0)
if os.environ.get("REFNERF_ZERO_WEIGHTS"):     # power experiment: the same instruction stream on all-zero operands
    _blob = _blob * 0.0
model.nerf_mlp.load_flat_params(_blob)
rays = utils.rays_from_dict(synthetic.blender_rays(R, seed=1, center_frac=0.5), dev)
ref = None
for prec in modes:
    cfg.hip_precision = prec
    n = 20 if prec == "f32" else 100
    with torch.no_grad():
        for _ in range(max(n // 4, 3)):
            out = model(rays, 1.0, True)
        torch.cuda.synchronize()
        _hip.set_timing(True) if hasattr(_hip, "set_timing") else None
        t0 = time.perf_counter()
        for _ in range(n):
            out = model(rays, 1.0, True)
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    rgb = out[0][-1]["rgb"].double()
    if prec == "f32":
        ref = rgb
    print(f"{prec:6s} {R}x{N}x2: {ms:.3f} ms/step = {R * N * 2 / ms * 1e3:.3e} ray-samples/s  rgb sum {float(rgb.sum()):.9f}", flush=True)
