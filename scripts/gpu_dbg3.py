import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import refnerf_pl_amd
from refnerf_pl_amd import _hip as hip, synthetic
from test_hip_parity import dev_rays
DEV = "cuda:0"
R = 3
P = synthetic.make_params(seed=3, bias_scale=0.05, sharpen=8.0)
rays = synthetic.blender_rays(R, seed=11, center_frac=0.5)
packed = hip.pack_weights(torch.tensor(P, device=DEV), precision=0)
r = dev_rays(rays)
sd = torch.tensor([[0.0, 1.0]], device=DEV).repeat(R, 1); w = torch.ones((R, 1), device=DEV)
for training in (0, 1):
    for prec in (0, 1):
        if training and prec: continue
        pk = hip.pack_weights(torch.tensor(P, device=DEV), precision=prec)
        cfg = hip.default_cfg(n_samples=33, n_in=1, precision=prec, training=training, opaque_background=1)
        res = hip.level_forward(pk, cfg, r, sd, w)
        torch.cuda.synchronize()
        bad = {k: int(torch.isnan(v).sum()) for k, v in res.items() if v.dtype.is_floating_point and torch.isnan(v).any()}
        print("training", training, "prec", prec, "nan:", bad, "acc", res["r_acc"].cpu().numpy(), "w_last", res["weights"][:, -1].cpu().numpy())
