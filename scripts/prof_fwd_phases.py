"""REFNERF_PROF cycle stamps of the training forward (mid-grid workgroup), f32 and bf16-chain mode."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import refnerf_pl_amd
from refnerf_pl_amd import _hip, synthetic
dev="cuda:0"; R,N=4096,128
P=torch.tensor(synthetic.make_params(0,0.05,20.0),device=dev)
rays={k: torch.tensor(v,device=dev) for k,v in synthetic.blender_rays(R,seed=1,center_frac=0.5).items()}
for k in ("radii","near","far"): rays[k]=rays[k].reshape(-1)
packed=_hip.pack_weights(P,precision=0)
sd=torch.tensor([[0.0,1.0]],device=dev).repeat(R,1); w=torch.ones((R,1),device=dev)
for prec in (0,1):
    cfg=_hip.default_cfg(n_samples=N,n_in=1,training=1,compute_extras=0)
    cfg.precision=prec
    _hip.level_forward(packed,cfg,rays,sd,w,history=True,save_activations=True)
    torch.cuda.synchronize()
    os.environ["REFNERF_PROF"]="1"
    print("precision",prec,file=sys.stderr)
    _hip.level_forward(packed,cfg,rays,sd,w,history=True,save_activations=True)
    torch.cuda.synchronize()
    del os.environ["REFNERF_PROF"]
