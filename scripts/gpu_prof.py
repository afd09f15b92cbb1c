import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refnerf_pl_amd
from refnerf_pl_amd import _hip, synthetic
from test_hip_parity import dev_rays
dev="cuda:0"
P = synthetic.make_params(0, 0.05, 20.0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rays = synthetic.blender_rays(R, seed=1, center_frac=0.5)
packed = _hip.pack_weights(torch.tensor(P, device=dev), precision=1)
r = dev_rays(rays)
sd = torch.tensor([[0.0, 1.0]], device=dev).repeat(R, 1); w = torch.ones((R, 1), device=dev)
cfg = _hip.default_cfg(n_samples=128, n_in=1, precision=1)
res = _hip.level_forward(packed, cfg, r, sd, w)
torch.cuda.synchronize()
os.environ["REFNERF_PROF"] = "1"
print("level 0", file=sys.stderr)
res = _hip.level_forward(packed, cfg, r, sd, w)
cfg1 = _hip.default_cfg(n_samples=128, n_in=128, precision=1)
print("level 1", file=sys.stderr)
res1 = _hip.level_forward(packed, cfg1, r, res["sdist"], res["weights"])
