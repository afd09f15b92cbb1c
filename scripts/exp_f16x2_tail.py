"""CPU experiment (no GPU): where does the tail of REFNERF_PREC_F16X2's RGB error on the harsher trained weight sets come
from?  Emulates the kernel's operand plan (scripts/exp_split_precision.py: spatial trunk + scalar heads = both operands
hi + lo; bottleneck = weights hi only, activations hi + lo; directional trunk + rgb = plain f16) on oracle/torch_path.py and
widens one group at a time.  VERDICT r03 "next" item 2.
  python scripts/exp_f16x2_tail.py [long|llff] [rays] [samples]
TEST / MEASUREMENT INFRASTRUCTURE (imports oracle/): never imported by the product."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import refnerf_pl_amd  # noqa: F401,E402
from refnerf_pl_amd import synthetic  # noqa: E402
import exp_split_precision as E  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "long"
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    torch.set_num_threads(8)
    kw = {}
    if which == "llff":
        blob = np.load(os.path.join(ROOT, "tests", "golden", "trained_llff_blob.npz"))["blob_f32"]
        rays = synthetic.llff_rays(n_rays, seed=3)
        kw = dict(srgb_mapping=False, render_srgb_mode="norm_linear")
    else:
        blob = np.load(os.path.join(ROOT, "tests", "golden", "trained_long_blob.npz"))["blob_f32"]
        rays = synthetic.blender_rays(n_rays, seed=3, center_frac=0.8)

    def run(modes):
        P = E.TP.unpack(blob)
        by_id = {}
        for g, names in E.GROUPS.items():
            for nme in names:
                by_id[id(P[nme][0])] = modes.get(nme, modes.get(g, "f32"))

        class FShim:
            def __getattr__(self, k):
                return getattr(torch.nn.functional, k)

            @staticmethod
            def linear(x, w, b=None):
                return E.emu_linear(x, w, b, by_id[id(w)])
        old, real_unpack = E.TP.F, E.TP.unpack
        E.TP.F, E.TP.unpack = FShim(), (lambda _b, _s=None: P)
        try:
            return E.TP.model_forward(blob, rays, num_prop_samples=N, num_nerf_samples=N, **kw)
        finally:
            E.TP.F, E.TP.unpack = old, real_unpack

    ref = run({})
    kernel = {"spatial": "f16x3", "heads": "f16x3", "bottleneck": "f16x2a", "dir": "f16", "rgb": "f16"}
    cases = [("kernel plan", kernel),
             ("+ bottleneck both split", dict(kernel, bottleneck="f16x3")),
             ("+ dir activations split", dict(kernel, dir="f16x2a")),
             ("+ dir weights split", dict(kernel, dir="f16x2w")),
             ("+ dir both split", dict(kernel, dir="f16x3")),
             ("+ rgb both split", dict(kernel, rgb="f16x3")),
             ("+ dir.0 both split", dict(kernel, **{"viewdir_mlp.0": "f16x3"})),
             ("+ dir + rgb + bottleneck all split (= all f16x3)", {g: "f16x3" for g in E.GROUPS}),
             ("spatial/heads f32, rest kernel plan", dict(kernel, spatial="f32", heads="f32")),
             ]
    for tag, m in cases:
        out = run(m)
        row = [f"{tag:52s}"]
        for lvl in range(2):
            e = np.abs(out[lvl]["r_rgb"] - ref[lvl]["r_rgb"]).max(-1)
            row.append("L%d max %.2e p99.9 %.2e idx %.5f" % (lvl, e.max(), np.quantile(e, 0.999),
                                                            np.mean(out[lvl]["bin_idx"] == ref[lvl]["bin_idx"])))
        print(" | ".join(row), flush=True)


if __name__ == "__main__":
    main()
