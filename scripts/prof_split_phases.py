"""REFNERF_PROF=1 cycle stamps of the split-f16 level kernel (mid-grid workgroup, last pass of the workgroup), C2 shape.
slots: 1 resample done, 2 first chunks landed; run 0: 3 IPE, 4 layer 0, 5 trunk, 6 heads; run 1: 7..10; directional phase:
11 head activations + IDE, 12 layer 0, 13 trunk, 14 rgb + colour + history; 15 passes done; 16 compositing"""
import os, sys
os.environ["REFNERF_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import refnerf_pl_amd
from refnerf_pl_amd import _hip, synthetic
from test_hip_parity import dev_rays
dev = "cuda:0"
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
P = synthetic.make_params(0, 0.05, 20.0)
R = int(os.environ.get("REFNERF_PROF_RAYS", "4096"))   # fewer rays = fewer busy CUs (R / 4 workgroups at 128 samples)
rays = synthetic.blender_rays(R, seed=1, center_frac=0.5)
packed = _hip.pack_weights(torch.tensor(P, device=dev), precision=prec)
r = dev_rays(rays)
sd = torch.tensor([[0.0, 1.0]], device=dev).repeat(R, 1); w = torch.ones((R, 1), device=dev)
cfg = _hip.default_cfg(n_samples=128, n_in=1, precision=prec)
for _ in range(3):
    res = _hip.level_forward(packed, cfg, r, sd, w)
torch.cuda.synchronize()
cfg1 = _hip.default_cfg(n_samples=128, n_in=128, precision=prec)
print("level 1", file=sys.stderr)
res1 = _hip.level_forward(packed, cfg1, r, res["sdist"], res["weights"])
