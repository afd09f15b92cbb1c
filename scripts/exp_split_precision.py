"""CPU experiment (no GPU): which GEMMs of the level need split (hi + lo) 16-bit operands to stay within 1e-4 RGB of
the fp32 evaluation on the trained-like weights?

Emulates the MFMA arithmetic modes of the level kernels with ATen ops on oracle/torch_path.py: operands rounded to the
16-bit type (optionally split into hi + lo with the lo*lo term dropped), products and sums in fp32.
  python scripts/exp_split_precision.py [rays] [samples]
TEST / MEASUREMENT INFRASTRUCTURE (imports oracle/): never imported by the product.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import refnerf_pl_amd  # noqa: F401,E402
from refnerf_pl_amd import synthetic  # noqa: E402
from oracle import torch_path as TP  # noqa: E402

real_linear = torch.nn.functional.linear


def split(x, dt, scale_lo=1.0):
    hi = x.to(dt).float()
    lo = ((x - hi) * scale_lo).to(dt).float() / scale_lo
    return hi, lo


def emu_linear(x, w, b, mode):
    if mode == "f32":
        return real_linear(x, w, b)
    dt = torch.float16 if "f16" in mode and not mode.startswith("bf") else torch.bfloat16
    if mode in ("f16", "bf16"):
        return real_linear(x.to(dt).float(), w.to(dt).float(), b)
    xh, xl = split(x, dt)
    wh, wl = split(w, dt, 2048.0 if mode.endswith("s") else 1.0)
    base = mode.rstrip("s")
    if base.endswith("x3"):      # hi*hi + lo*hi + hi*lo
        return real_linear(xh, wh, b) + real_linear(xl, wh) + real_linear(xh, wl)
    if base.endswith("x2w"):     # weights split only
        return real_linear(xh, wh, b) + real_linear(xh, wl)
    if base.endswith("x2a"):     # activations split only
        return real_linear(xh, wh, b) + real_linear(xl, wh)
    raise ValueError(mode)


GROUPS = {"spatial": ["spatial_net.%d" % i for i in range(8)],
          "heads": ["raw_density", "grad_pred", "raw_roughness", "raw_rgb_diffuse", "raw_tint", "bottleneck"],
          "dir": ["viewdir_mlp.%d" % i for i in range(8)],
          "rgb": ["rgb"]}


def run(blob, rays, N, modes):
    """modes: {group or layer name: mode}"""
    P = TP.unpack(blob)
    by_id = {}
    for g, names in GROUPS.items():
        for nme in names:
            by_id[id(P[nme][0])] = modes.get(nme, modes.get(g, "f32"))

    class FShim:
        def __getattr__(self, k):
            return getattr(torch.nn.functional, k)

        @staticmethod
        def linear(x, w, b=None):
            return emu_linear(x, w, b, by_id[id(w)])
    old = TP.F
    TP.F = FShim()
    real_unpack = TP.unpack
    TP.unpack = lambda _b, _s=None: P
    try:
        out = TP.model_forward(blob, rays, num_prop_samples=N, num_nerf_samples=N)
    finally:
        TP.F = old
        TP.unpack = real_unpack
    return out


def report(tag, out, ref):
    row = [tag]
    for lvl in range(len(ref)):
        e = np.abs(out[lvl]["r_rgb"] - ref[lvl]["r_rgb"]).max()
        agree = np.mean(out[lvl]["bin_idx"] == ref[lvl]["bin_idx"])
        row.append("L%d rgb %.2e w %.2e idx %.4f" % (lvl, e, np.abs(out[lvl]["weights"] - ref[lvl]["weights"]).max(), agree))
    print(" | ".join(row), flush=True)


def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    torch.set_num_threads(8)
    blob = np.load(os.path.join(ROOT, "tests", "golden", "trained_blob.npz"))["blob_f16"].astype(np.float32)
    rays = synthetic.blender_rays(n_rays, seed=3, center_frac=0.8)
    ref = run(blob, rays, N, {})
    allg = lambda m: {g: m for g in GROUPS}
    cases = [("all f16", allg("f16")), ("all bf16", allg("bf16")),
             ("all f16x3", allg("f16x3")), ("all f16x3s (lo*2^11)", allg("f16x3s")), ("all bf16x3", allg("bf16x3")),
             ("all f16x2w", allg("f16x2w")), ("all f16x2a", allg("f16x2a")),
             ("spatial+heads f16x3, dir+rgb f16", {"spatial": "f16x3", "heads": "f16x3", "dir": "f16", "rgb": "f16"}),
             ("spatial+heads f16x3, dir+rgb bf16", {"spatial": "f16x3", "heads": "f16x3", "dir": "bf16", "rgb": "bf16"}),
             ("spatial f16, heads+dir+rgb f16x3", {"spatial": "f16", "heads": "f16x3", "dir": "f16x3", "rgb": "f16x3"}),
             ("spatial+heads f16x3, dir f16x2w", {"spatial": "f16x3", "heads": "f16x3", "dir": "f16x2w", "rgb": "f16x2w"}),
             ("spatial+heads f16x3, dir f16x2a", {"spatial": "f16x3", "heads": "f16x3", "dir": "f16x2a", "rgb": "f16x2a"}),
             ]
    for i in range(8):
        m = allg("f16x3")
        m["spatial_net.%d" % i] = "f16"
        cases.append(("f16x3 except spatial_net.%d f16" % i, m))
    for nme in GROUPS["heads"]:
        m = allg("f16x3")
        m[nme] = "f16"
        cases.append(("f16x3 except %s f16" % nme, m))
    for tag, m in cases:
        report(tag, run(blob, rays, N, m), ref)


if __name__ == "__main__":
    main()
