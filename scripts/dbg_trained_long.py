import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from helpers import load_golden, params_from_golden, rays_from_golden
import refnerf_pl_amd
from refnerf_pl_amd import configs, models, train_utils, utils, layout
g = load_golden("model_trained_long_train")
ref_sub = g["grads_sub"]; tn = g["grads_tensor_l2"]
out = {}
for fwd, bwd in (("f32","f32"),("f16x2","f32"),("f32","f16x2"),("f16x2","f16x2")):
    configs.clear_config()
    configs.parse_config_files_and_bindings(["configs/refnerf_blender.gin"], [str(b) for b in g["bindings"] if str(b)] + [f"Config.hip_train_precision = '{fwd}'", f"Config.hip_bwd_precision = '{bwd}'"])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to("cuda:0").train()
    model.nerf_mlp.load_flat_params(params_from_golden(g))
    rays = utils.rays_from_dict(rays_from_golden(g), "cuda:0")
    batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
    rend, hist = model(rays, 1.0, False)
    total, terms, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
    total.backward()
    grads = torch.cat([p.grad.flatten() for p in model.nerf_mlp.ordered_parameters()]).cpu().numpy()
    out[(fwd,bwd)] = grads
    rel = np.linalg.norm(grads[::97]-ref_sub)/np.linalg.norm(ref_sub)
    print(fwd, bwd, "rel vs reference %.2e" % rel, "normals diff vs", end=" ")
    print("terms", {k: float(v.detach()) for k, v in terms.items()})
base = out[("f32","f32")]
for k, gr in out.items():
    d = gr - base
    print(k, "rel vs f32/f32 %.2e" % (np.linalg.norm(d)/np.linalg.norm(base)))
    worst = []
    for s in layout.PARAM_SPECS:
        a = slice(s.w_off, s.w_off + s.out_dim*s.in_dim)
        worst.append((np.linalg.norm(d[a])/max(np.linalg.norm(base[a]),1e-30), s.name))
    print("   worst tensors", sorted(worst)[-5:])
