"""CPU experiment in the BUILD CONTAINER (imports the reference from /root/reference through tests/golden/_ref_harness.py;
it never travels): how many bits do the operands of the weight-gradient contraction dW = DELTA^T x ACT need?
VERDICT r03 "next" item 4, step 1.  The reference's own forward, losses and autograd run unchanged, except that every
nn.Linear computes its WEIGHT / BIAS gradient from rounded copies of its two operands (the input gradient stays exact: that
is the chain kernels' business):
  ACT   = the layer input  (rows the training forward saves)      -- 'f32' | 'bf16x2' (hi + lo bf16: today's GEMM) | 'f16x2' | 'f16' | 'bf16'
  DELTA = the pre-activation gradient (rows the backward saves)   -- 'f32' | 'bf16x2' | 'bf16' | 'f16s' (ONE f16 after a per-sample
          power-of-two scale: 11 bits, what level_bwd_f16x2c already holds as its hi halves) | 'f16x2s'
Reported: rel-L2 of the whole gradient against the unrounded autograd, on the trained weight sets' training fixtures.
  python scripts/exp_delta_precision.py
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

MODE = {"act": "f32", "delta": "f32"}


def rnd(x, how):
    if how == "f32":
        return x
    if how == "bf16":
        return x.to(torch.bfloat16).float()
    if how == "f16":
        return x.to(torch.float16).float()
    if how == "bf16x2":
        hi = x.to(torch.bfloat16).float()
        return hi + (x - hi).to(torch.bfloat16).float()
    if how == "f16x2":
        hi = x.to(torch.float16).float()
        return hi + (x - hi).to(torch.float16).float()
    if how in ("f16s", "f16x2s"):        # per-sample (row) power-of-two scale so that the row's largest entry sits at ~2^12
        m = x.abs().amax(dim=-1, keepdim=True)
        s = torch.where(m > 1e-30, torch.exp2(12.0 - torch.ceil(torch.log2(m.clamp_min(1e-30)))), torch.ones_like(m))
        y = x * s
        hi = y.to(torch.float16).float()
        if how == "f16x2s":
            hi = hi + (y - hi).to(torch.float16).float()
        return hi / s
    if how.startswith("f16g"):           # what the kernels do: per-sample factor to 2^7..2^8, ONE half, then every sample brought to
        top = int(how[4:] or 8)          # the layer's smallest factor (largest deltas at 2^top): small samples shift down / underflow
        m = x.abs().amax(dim=-1, keepdim=True)
        s = torch.where(m > 1e-30, torch.exp2(8.0 - torch.ceil(torch.log2(m.clamp_min(1e-30)))), torch.ones_like(m))   # row max -> [2^7, 2^8)
        d16 = (x * s).to(torch.float16)
        smin = s[m > 1e-30].min() if bool((m > 1e-30).any()) else torch.tensor(1.0)
        f = torch.where(m > 1e-30, (smin / s) * (2.0 ** (top - 8)), torch.zeros_like(s))
        shifted = (d16 * f.to(torch.float16)).float()       # half x half product, rounded to half (subnormals kept)
        return shifted / (smin * (2.0 ** (top - 8)))
    raise ValueError(how)


class LinearRoundedWgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gy @ w
        a = rnd(x.reshape(-1, x.shape[-1]), MODE["act"])
        d = rnd(gy.reshape(-1, gy.shape[-1]), MODE["delta"])
        return gx, d.t() @ a, d.sum(0)


def patched_forward(self, x):
    return LinearRoundedWgrad.apply(x, self.weight, self.bias)


def grads_of(model, cfg, rays, gt):
    return MG.run_model(model, cfg, rays, True, gt)["grads"]


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
    combos = [("bf16x2", "bf16x2"), ("f16x2", "f16s"), ("f16x2", "f16g8"), ("f16x2", "f16g15")]
    real = torch.nn.Linear.forward
    for name, (blob, bindings, rays, target) in sets.items():
        model, cfg = MG.build_model_blob(bindings, blob)
        gt = target(rays)
        MODE.update(act="f32", delta="f32")
        exact = grads_of(model, cfg, rays, gt)
        torch.nn.Linear.forward = patched_forward
        try:
            MODE.update(act="f32", delta="f32")
            same = grads_of(model, cfg, rays, gt)
            print(f"{name}: |g| = {np.linalg.norm(exact):.4g}; patched-but-unrounded vs autograd {np.linalg.norm(same - exact) / np.linalg.norm(exact):.1e}")
            for act, delta in combos:
                MODE.update(act=act, delta=delta)
                g = grads_of(model, cfg, rays, gt)
                rel = np.linalg.norm(g - exact) / np.linalg.norm(exact)
                worst = max(np.linalg.norm((g - exact)[s.w_off:s.w_off + s.out_dim * s.in_dim]) / max(np.linalg.norm(exact[s.w_off:s.w_off + s.out_dim * s.in_dim]), 1e-30)
                            for s in MG.layout.PARAM_SPECS)
                print(f"   ACT {act:7s} x DELTA {delta:7s}: gradient rel-L2 {rel:.2e}   worst tensor {worst:.2e}", flush=True)
        finally:
            torch.nn.Linear.forward = real


if __name__ == "__main__":
    main()
