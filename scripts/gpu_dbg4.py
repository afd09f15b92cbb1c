import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import refnerf_pl_amd
from refnerf_pl_amd import _hip as hip, synthetic
from test_hip_parity import dev_rays
from oracle import oracle as O
DEV = "cuda:0"
R = 16
P = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0)
rays = synthetic.blender_rays(R, seed=6, center_frac=0.4)
r = dev_rays(rays)
sd = torch.tensor([[0.0, 1.0]], device=DEV).repeat(R, 1); w = torch.ones((R, 1), device=DEV)
pk = hip.pack_weights(torch.tensor(P, device=DEV), precision=0)
ref = O.model_forward(P, rays, training=1, num_levels=1, num_nerf_samples=128)
for training in (0, 1):
    cfg = hip.default_cfg(n_samples=128, n_in=1, precision=0, training=training)
    res = hip.level_forward(pk, cfg, r, sd, w)
    torch.cuda.synchronize()
    print("training", training, "weights err", np.abs(res["weights"].cpu().numpy() - ref[0]["weights"]).max(), "density err", np.abs(res["density"].cpu().numpy() - ref[0]["density"]).max(),
          "acc", res["r_acc"][:4].cpu().numpy(), ref[0]["r_acc"][:4])
cfg = hip.default_cfg(n_samples=128, n_in=1, precision=0, training=1)
res = hip.level_forward(pk, cfg, r, sd, w)
torch.cuda.synchronize()
for k in ("sdist", "density", "rgb", "normals", "normals_pred", "roughness", "weights", "r_distance", "r_rgb"):
    a = res[k].cpu().numpy(); b = ref[0][k].reshape(a.shape)
    print(k, "max err", np.abs(a - b).max(), "first", a.reshape(-1)[:4], b.reshape(-1)[:4])
