"""GPU debug aid for the round-5 training kernels of the f16x2 mode (refnerf_sq_train.hip): one training step of the smoke
model in the f32 chains (the strict-parity kernels) and in the f16x2 chains, compared output by output and gradient tensor by
gradient tensor, plus the oracle's totals.  `REFNERF_LEGACY_F16X2_TRAIN=1 python scripts/dbg_sq_train.py` runs the round-4
kernels instead.  MEASUREMENT / DEBUG INFRASTRUCTURE: never imported by the product."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import refnerf_pl_amd  # noqa: F401,E402
from refnerf_pl_amd import _hip, configs, layout, models, synthetic, train_utils, utils  # noqa: E402


def run(model, cfg, rays_t, gt, chains):
    cfg.hip_train_precision = cfg.hip_bwd_precision = chains
    for prm in model.parameters():
        prm.grad = None
    renderings, history = model(rays_t, 1.0, True)
    total, terms, _ = train_utils.compute_losses(model, utils.Batch(rays=rays_t, rgb=gt), rays_t, renderings, history, cfg)
    total.backward()
    torch.cuda.synchronize()
    grads = np.zeros(layout.NUM_PARAMS, np.float32)
    per = {}
    for spec, lin in model.nerf_mlp._named_linears():
        w = lin.weight.grad.reshape(-1).cpu().numpy()
        b = lin.bias.grad.cpu().numpy()
        grads[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = w
        grads[spec.b_off:spec.b_off + spec.out_dim] = b
        per[spec.name] = (w.copy(), b.copy())
    outs = {}
    for lvl in range(len(renderings)):
        for k, v in renderings[lvl].items():
            if torch.is_tensor(v):
                outs[f"L{lvl}.r.{k}"] = v.detach().float().cpu().numpy()
        for k, v in history[lvl].items():
            if torch.is_tensor(v):
                outs[f"L{lvl}.h.{k}"] = v.detach().float().cpu().numpy()
    return float(total.detach()), grads, per, outs


def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    _hip.require_device()
    dev = torch.device("cuda:0")
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(ROOT, "configs", "refnerf_blender.gin")],
                                            [f"Model.num_prop_samples = {ns}", f"Model.num_nerf_samples = {ns}"])
    cfg = configs.Config()
    model = models.construct_model(None, cfg).to(dev).train()
    blob = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0)
    model.nerf_mlp.load_flat_params(blob)
    rays_np = synthetic.blender_rays(n_rays, seed=1, center_frac=0.4)
    gt = synthetic.target_rgb(n_rays, seed=5)
    rays_t = utils.rays_from_dict(dict(rays_np), dev)
    t32, g32, p32, o32 = run(model, cfg, rays_t, gt, "f32")
    t16, g16, p16, o16 = run(model, cfg, rays_t, gt, "f16x2")
    print(f"legacy={_hip.LEGACY_F16X2_TRAIN}  total loss f32 {t32:.8f}  f16x2 {t16:.8f}")
    for k in sorted(o32):
        a, b = o32[k], o16[k]
        if a.shape != b.shape:
            print(f"  {k}: shape {a.shape} vs {b.shape}")
            continue
        err = np.abs(a - b)
        bad = ~np.isfinite(b)
        print(f"  {k:28s} max|diff| {np.nanmax(err) if err.size else 0:.3e}  (|ref| max {np.abs(a).max() if a.size else 0:.3e})  nonfinite {int(bad.sum())}")
    print(f"gradient rel-L2 f16x2 vs f32 chains: {np.linalg.norm(g16 - g32) / np.linalg.norm(g32):.3e}   |g| {np.linalg.norm(g32):.4e}")
    for name in p32:
        for i, part in enumerate(("w", "b")):
            a, b = p32[name][i], p16[name][i]
            na = np.linalg.norm(a)
            rel = np.linalg.norm(a - b) / max(na, 1e-30)
            print(f"  {name + '.' + part:34s} |g| {na:.3e}  rel {rel:.3e}  nonfinite {int((~np.isfinite(b)).sum())}")
    np.set_printoptions(linewidth=200, precision=4)
    for nm in ("bottleneck",):
        a, b = p32[nm][1], p16[nm][1]
        print(nm, "bias grad f32  :", a[:24])
        print(nm, "bias grad f16x2:", b[:24])
        print(nm, "ratio          :", (b / a)[:48])
        aw, bw = p32[nm][0].reshape(128, 256), p16[nm][0].reshape(128, 256)
        print(nm, "per-row rel err of dW:", (np.linalg.norm(aw - bw, axis=1) / np.linalg.norm(aw, axis=1))[:48])
    try:
        from oracle import oracle as O
        tr = {k: v for k, v in rays_np.items()}
        o_losses, o_grads, _ = O.model_train(blob, tr, gt, num_prop_samples=ns, num_nerf_samples=ns)
        print(f"oracle total {o_losses['total']:.8f}; grads rel-L2 vs oracle: f32 {np.linalg.norm(g32 - o_grads) / np.linalg.norm(o_grads):.3e}  "
              f"f16x2 {np.linalg.norm(g16 - o_grads) / np.linalg.norm(o_grads):.3e}")
    except Exception as e:  # noqa: BLE001
        print("oracle comparison skipped:", e)


if __name__ == "__main__":
    main()
