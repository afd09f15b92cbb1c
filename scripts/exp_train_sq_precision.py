"""CPU experiment in the BUILD CONTAINER (imports the reference from /root/reference through tests/golden/_ref_harness.py;
it never travels): what does the arithmetic of the round-5 training kernels (the eval kernel's skeleton) cost in gradient
accuracy?  The reference's forward, losses and autograd run unchanged except inside nn.Linear:

  forward   spatial layers + scalar heads : operands at 22 bits (hi + lo halves)                      -- as today
            bottleneck                    : W at 11 bits (hi only), x at 22 bits                      -- the eval kernel's SQ_BN
            directional layers + rgb      : W and x at 11 bits (plain f16)                            -- the eval kernel's trunk
  dX        spatial                       : 22-bit operands (three products)
            directional                   : 'x3' three products (22 bits) | 'x1' plain f16 (delta: one half after the
                                            per-sample power-of-two factor, W^T: 11 bits)
  dW        ACT spatial 22 bits, ACT directional = the forward's own 11-bit input (nothing dropped: that IS what the forward
            multiplied), DELTA one half after the per-sample factor (today's format)

Reported: rel-L2 of the whole gradient against the reference's unmodified autograd, per trained weight set.
  python scripts/exp_train_sq_precision.py
MEASUREMENT INFRASTRUCTURE: never imported by the product."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG  # noqa: E402  (brings the reference in through the harness)
from refnerf_pl_amd import synthetic  # noqa: E402
from exp_delta_precision import rnd  # noqa: E402

CFG = {"fwd_sp": "f16x2", "fwd_dir": "f16", "fwd_dir_w": None, "dx_dir": "x3", "act_dir": "f16", "delta": "f16g8", "act_sp": "f16x2", "on": False}
KIND = {}


class LinearSq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, kind):
        if kind == "dir":
            xe, we = rnd(x, CFG["fwd_dir"]), rnd(w, CFG["fwd_dir_w"] or CFG["fwd_dir"])
        elif kind == "bneck":
            xe, we = rnd(x, CFG["fwd_sp"]), rnd(w, CFG["bneck_w"] if "bneck_w" in CFG else "f16")
        else:
            xe, we = rnd(x, CFG["fwd_sp"]), rnd(w, CFG["fwd_sp"])
        ctx.kind = kind
        ctx.save_for_backward(xe, we)
        return torch.nn.functional.linear(xe, we, b)

    @staticmethod
    def backward(ctx, gy):
        xe, we = ctx.saved_tensors
        kind = ctx.kind
        g2 = gy.reshape(-1, gy.shape[-1])
        if (kind == "dir" and CFG["dx_dir"] == "x1") or (kind != "dir" and CFG.get("dx_sp") == "x2"):
            gx = rnd(g2, "f16s") @ we                       # delta: one half per element after the per-sample factor; W^T as the forward had it
        else:
            gx = (rnd(g2, "f16x2s") if CFG["fwd_sp"] != "f32" else g2) @ we
        a = xe.reshape(-1, xe.shape[-1])
        a = rnd(a, CFG["act_dir"] if kind == "dir" else CFG["act_sp"])
        d = rnd(g2, CFG["delta"])
        return gx.reshape(xe.shape), d.t() @ a, d.sum(0), None


def patched_forward(self, x):
    if not CFG["on"]:
        return torch.nn.functional.linear(x, self.weight, self.bias)
    return LinearSq.apply(x, self.weight, self.bias, KIND.get(id(self), "sp"))


def tag_layers(model):
    KIND.clear()
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.Linear):
            if "viewdir_mlp" in name or name.endswith(".rgb"):
                KIND[id(mod)] = "dir"
            elif name.endswith("bottleneck"):
                KIND[id(mod)] = "bneck"
            else:
                KIND[id(mod)] = "sp"


def main():
    torch.set_num_threads(8)
    sets = {
        "trained (400 steps)": (MG._load_trained_blob(), ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 96"],
                                synthetic.blender_rays(64, seed=32, center_frac=0.8), MG.analytic_target),
        "trained_long (2500 steps)": (np.load(MG.TRAINED_LONG_BLOB)["blob_f32"], ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 96"],
                                      synthetic.blender_rays(16, seed=42, center_frac=0.8), MG.analytic_target),
        "trained_llff (1200 steps)": (np.load(MG.TRAINED_LLFF_BLOB)["blob_f32"], MG.LLFF_BINDINGS + ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 96"],
                                      synthetic.llff_rays(64, seed=43), MG.analytic_target_ndc),
    }
    base = dict(fwd_sp="f16x2", fwd_dir="f16x2", fwd_dir_w=None, dx_dir="x3", dx_sp="x3", act_dir="f16x2", act_sp="f16x2", delta="f16g8", bneck_w="f16")
    new = dict(fwd_dir="f16", fwd_dir_w="f16x2", act_dir="f16", bneck_w="f16x2", dx_dir="x1")
    combos = [
        ("today: all 22-bit chains", dict()),
        ("r5: dirW22xX11, bneckW22, dXdir 2p", dict(new)),
        ("r5 + dX spatial 2 products", dict(new, dx_sp="x2")),
        ("r5 + dX sp 2p + ACT_SP 11 bits", dict(new, dx_sp="x2", act_sp="f16")),
        ("r5 + dX sp 2p, exact dW operands", dict(new, dx_sp="x2", act_sp="f32", act_dir="f32", delta="f32")),
        ("today, exact dW operands", dict(act_sp="f32", act_dir="f32", delta="f32")),
    ]
    real = torch.nn.Linear.forward
    for name, (blob, bindings, rays, target) in sets.items():
        model, cfg = MG.build_model_blob(bindings, blob)
        gt = target(rays)
        CFG["on"] = False
        exact = MG.run_model(model, cfg, rays, True, gt)["grads"]
        tag_layers(model)
        torch.nn.Linear.forward = patched_forward
        try:
            print(f"{name}: |g| = {np.linalg.norm(exact):.4g}")
            for label, kw in combos:
                CFG.update(base)
                CFG.update(kw)
                CFG["on"] = True
                g = MG.run_model(model, cfg, rays, True, gt)["grads"]
                CFG["on"] = False
                rel = np.linalg.norm(g - exact) / np.linalg.norm(exact)
                worst = max(np.linalg.norm((g - exact)[s.w_off:s.w_off + s.out_dim * s.in_dim]) / max(np.linalg.norm(exact[s.w_off:s.w_off + s.out_dim * s.in_dim]), 1e-30)
                            for s in MG.layout.PARAM_SPECS)
                print(f"   {label:32s}: gradient rel-L2 {rel:.2e}   worst tensor {worst:.2e}", flush=True)
        finally:
            torch.nn.Linear.forward = real


if __name__ == "__main__":
    main()
