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
g_rgb=torch.randn((R,3),device=dev)*1e-3; g_w=torch.randn((R,N),device=dev)*1e-3; g_np=torch.randn((R,N,3),device=dev)*1e-3
for prec in (0,1):
    cfg=_hip.default_cfg(n_samples=N,n_in=1,training=1,compute_extras=0)
    cfg.precision=prec                       # the forward in the same mode (bf16: bf16 ACT rows + sample-major block)
    res=_hip.level_forward(packed,cfg,rays,sd,w,history=True,save_activations=True)
    grads=torch.zeros(_hip.NUM_PARAMS,device=dev)
    _hip.level_backward(packed,cfg,rays,res,g_rgb,g_w,g_np,grads)
    torch.cuda.synchronize()
    os.environ["REFNERF_PROF"]="1"
    print("precision",prec,file=sys.stderr)
    _hip.level_backward(packed,cfg,rays,res,g_rgb,g_w,g_np,grads)
    torch.cuda.synchronize()
    del os.environ["REFNERF_PROF"]
