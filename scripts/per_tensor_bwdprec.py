import os, sys
sys.path.insert(0, os.getcwd())
import torch
import refnerf_pl_amd
from refnerf_pl_amd import _hip, synthetic, layout
dev="cuda:0"; R,N=64,128
P=torch.tensor(synthetic.make_params(0,0.05,20.0),device=dev)
rays={k: torch.tensor(v,device=dev) for k,v in synthetic.blender_rays(R,seed=1,center_frac=0.5).items()}
for k in ("radii","near","far"): rays[k]=rays[k].reshape(-1)
packed=_hip.pack_weights(P,precision=0)
sd=torch.tensor([[0.0,1.0]],device=dev).repeat(R,1); w=torch.ones((R,1),device=dev)
g=torch.Generator().manual_seed(0)
g_rgb=(torch.randn((R,3),generator=g)*1e-2).to(dev); g_w=(torch.randn((R,N),generator=g)*1e-2).to(dev); g_np=(torch.randn((R,N,3),generator=g)*1e-2).to(dev)
out={}
for prec in (0,1):
    cfg=_hip.default_cfg(n_samples=N,n_in=1,training=1,compute_extras=0)
    res=_hip.level_forward(packed,cfg,rays,sd,w,history=True,save_activations=True)
    cfg.precision=prec
    grads=torch.zeros(_hip.NUM_PARAMS,device=dev)
    _hip.level_backward(packed,cfg,rays,res,g_rgb,g_w,g_np,grads)
    out[prec]=grads.double().cpu()
a,b=out[0],out[1]
for s in layout.PARAM_SPECS:
    sl=slice(s.w_off,s.w_off+s.out_dim*s.in_dim); bl=slice(s.b_off,s.b_off+s.out_dim)
    print(f"{s.name:18s} w rel {float((a[sl]-b[sl]).norm()/a[sl].norm()):.2e}  b rel {float((a[bl]-b[bl]).norm()/max(float(a[bl].norm()),1e-30)):.2e}")
