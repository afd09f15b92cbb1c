"""TEST INFRASTRUCTURE: the CPU oracle behind the Model.__call__ call surface, with autograd.

`OracleModel(params, ...)` returns (renderings, ray_history) as torch (CPU) tensors whose gradients
flow into `OracleModel.grads` through rn_level_backward, so that the SAME loss code
(refnerf_pl_amd.train_utils) can run on top of the oracle and on top of the HIP path.
"""
import numpy as np
import torch

from oracle import oracle as O

_RAY = ("r_rgb", "r_diffuse", "r_specular", "r_acc", "r_distance", "r_normals", "r_normals_pred", "r_tint",
        "r_roughness")
_SAMPLE = ("weights", "density", "roughness", "rgb", "normals_pred", "tint", "diffuse", "specular")
_DIFF = _RAY + _SAMPLE


class _OracleLevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, cfg, rays, sdist_in, weights_in, holder):
        params, grads = holder["params"], holder["grads"]
        res = O.level_forward(params, cfg, rays, sdist_in.numpy(), weights_in.numpy(), n_threads=owner.n_threads)
        ctx.owner, ctx.cfg, ctx.rays, ctx.params, ctx.grads = owner, cfg, rays, params, grads
        ctx.sd_in, ctx.w_in = sdist_in.numpy().copy(), weights_in.numpy().copy()
        keys = tuple(k for k in _DIFF if k in res) + tuple(k for k in res if k not in _DIFF)
        holder["keys"] = keys
        ctx.n_diff = sum(1 for k in keys if k in _DIFF)
        ctx.keys = keys
        outs = tuple(torch.from_numpy(np.ascontiguousarray(res[k])) for k in keys)
        ctx.mark_non_differentiable(*outs[ctx.n_diff:])
        return outs

    @staticmethod
    def backward(ctx, *gs):
        seeds = {k: (None if g is None else g.detach().numpy()) for k, g in zip(ctx.keys[:ctx.n_diff], gs)}
        O.level_backward(ctx.params, ctx.cfg, ctx.rays, ctx.sd_in, ctx.w_in, seeds, grads=ctx.grads,
                         n_threads=ctx.owner.n_threads)
        return (None,) * 6


class OracleModel:
    """Model.__call__ (models.py:129-321) over oracle.level_forward / level_backward (training mode)."""
    def __init__(self, params, num_levels=2, num_prop_samples=128, num_nerf_samples=128, n_threads=0, vis_num_rays=16,
                 prop_params=None, prop_cfg_kw=None, **cfg_kw):
        self.params = np.ascontiguousarray(params, np.float32)
        self.grads = np.zeros_like(self.params)
        # Model.single_mlp = False: a separate proposal network (same architecture) for the levels before the last
        self.single_mlp = prop_params is None
        self.prop_params = self.params if prop_params is None else np.ascontiguousarray(prop_params, np.float32)
        self.prop_grads = self.grads if prop_params is None else np.zeros_like(self.prop_params)
        self.prop_cfg_kw = prop_cfg_kw or {}
        self.num_levels, self.num_prop_samples, self.num_nerf_samples = num_levels, num_prop_samples, num_nerf_samples
        self.n_threads, self.cfg_kw, self.vis_num_rays = n_threads, cfg_kw, vis_num_rays
        self._anchor = torch.zeros((), requires_grad=True)     # makes the level nodes part of the graph

    def __call__(self, rays, train_frac, compute_extras):
        del train_frac
        rd = {k: np.asarray(torch.as_tensor(getattr(rays, k)).detach().cpu().numpy(), np.float32)
              for k in ("origins", "directions", "viewdirs", "radii", "near", "far")}
        for k in ("radii", "near", "far"):
            rd[k] = rd[k].reshape(-1)
        R = rd["origins"].shape[0]
        sdist = torch.tensor([[self.cfg_kw.get("s_near", 0.0), self.cfg_kw.get("s_far", 1.0)]]).repeat(R, 1)
        weights = torch.ones((R, 1))
        renderings, history = [], []
        for lvl in range(self.num_levels):
            n = self.num_prop_samples if lvl < self.num_levels - 1 else self.num_nerf_samples
            is_prop = lvl < self.num_levels - 1
            kw = dict(self.cfg_kw, **(self.prop_cfg_kw if is_prop else {}))
            cfg = O.default_cfg(n_samples=n, n_in=weights.shape[1], training=1, compute_extras=int(bool(compute_extras)), **kw)
            holder = {"params": self.prop_params if is_prop else self.params, "grads": self.prop_grads if is_prop else self.grads}
            outs = _OracleLevel.apply(self, cfg, rd, sdist.detach() + 0 * self._anchor, weights.detach(), holder)
            res = dict(zip(holder["keys"], outs))
            sdist, weights = res["sdist"], res["weights"]
            rend = {"rgb": res["r_rgb"], "diffuse": res["r_diffuse"], "specular": res["r_specular"],
                    "distance": res["r_distance"][:, None], "acc": res["r_acc"]}
            if compute_extras:
                rend.update(normals=res["r_normals"], normals_pred=res["r_normals_pred"], tint=res["r_tint"],
                            roughness=res["r_roughness"][:, None], distance_mean=res["r_distance_mean"])
            renderings.append(rend)
            history.append({k: res[k] for k in ("density", "rgb", "normals", "normals_pred", "grad_pred", "tint",
                                                "diffuse", "specular", "sdist", "weights")}
                           | {"roughness": res["roughness"][..., None]})
        return renderings, history
