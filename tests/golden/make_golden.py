#!/usr/bin/env python3
"""Capture golden vectors from the upstream reference (build container only).

Run:  python tests/golden/make_golden.py
Needs /root/reference (read-only).  Writes tests/golden/*.npz: INPUTS and the
reference's OUTPUTS only -- no reference source travels.  Weights are not
stored: every fixture records the (seed, bias_scale, sharpen, roughness_bias)
arguments of refnerf_pl_amd.synthetic.make_params that regenerate them.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import _ref_harness  # noqa: E402

_ref_harness.install()
import torch  # noqa: E402
import gin  # noqa: E402

import refnerf_pl_amd  # noqa: E402,F401
from refnerf_pl_amd import layout, synthetic  # noqa: E402

from internal import configs, coord, models, ref_utils, render, stepfun, train_utils, utils  # noqa: E402

torch.set_num_threads(8)
REF_CFG = os.path.join(_ref_harness.REFERENCE_ROOT, "configs", "blender_refnerf.gin")


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def to_rays(d):
    return utils.Rays(**{k: torch.tensor(v) for k, v in d.items()})


def build_model(bindings=(), param_kw=None):
    gin.clear_config()
    gin.parse_config_files_and_bindings([REF_CFG], list(bindings))
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg)
    blob = synthetic.make_params(**(param_kw or {}))
    sd = model.nerf_mlp.state_dict()
    for spec in layout.PARAM_SPECS:
        w = blob[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim].reshape(spec.out_dim, spec.in_dim)
        b = blob[spec.b_off:spec.b_off + spec.out_dim]
        assert tuple(sd[spec.name + ".weight"].shape) == w.shape, spec
        sd[spec.name + ".weight"].copy_(torch.tensor(w))
        sd[spec.name + ".bias"].copy_(torch.tensor(b))
    return model, cfg


# --------------------------------------------------------------------------
def golden_sampler():
    out = {}
    rng = np.random.default_rng(7)
    cases = []
    # (a) level-0 degenerate step function
    cases.append((np.tile(np.array([[0.0, 1.0]], np.float32), (3, 1)), np.ones((3, 1), np.float32), 128))
    cases.append((np.tile(np.array([[0.0, 1.0]], np.float32), (2, 1)), np.ones((2, 1), np.float32), 64))
    # (b) peaky weights on a 128-bin histogram (what level 1 sees)
    t = np.sort(rng.random((6, 129)).astype(np.float32), axis=-1)
    t[:, 0] = 0.0
    t[:, -1] = 1.0
    w = (rng.random((6, 128)) ** 8).astype(np.float32)
    w[0, 40:44] = 5.0
    w[1, :] = 1e-6
    w[1, 100] = 0.9
    cases.append((t, w, 128))
    cases.append((t[:, ::2][:, :33].copy(), w[:, :32].copy(), 48))
    # (c) zero-width intervals and exact-zero weights
    t2 = t.copy()
    t2[:, 10:20] = t2[:, 10:11]
    w2 = w.copy()
    w2[:, 60:70] = 0.0
    cases.append((t2, w2, 192))
    # (d) uniform weights, exact binary fractions: ties u == cw happen here
    t3 = np.tile(np.linspace(0, 1, 65, dtype=np.float32), (2, 1))
    w3 = np.full((2, 64), 1.0 / 64, np.float32)
    cases.append((t3, w3, 64))
    for i, (t, w, n) in enumerate(cases):
        tt, ww = torch.tensor(t), torch.tensor(w)
        logits = torch.where(tt[..., 1:] > tt[..., :-1], 1.0 * torch.log(ww + 0.01), -float("inf"))
        sd = stepfun.sample_intervals(tt, logits, n, single_jitter=False, domain=(0.0, 1.0), use_gpu_resampling=False)
        # bin index implied by math.sorted_interp's mask (math.py:93)
        eps = torch.finfo(torch.float32).eps
        pad = 1 / (2 * n)
        u = torch.linspace(pad, 1.0 - pad - eps, n)
        cw = stepfun.integrate_weights(torch.softmax(logits, dim=-1))
        idx = (u[None, None, :] >= cw[:, :, None]).sum(dim=1) - 1
        out[f"c{i}_t"], out[f"c{i}_w"], out[f"c{i}_n"] = t, w, n
        out[f"c{i}_logits"], out[f"c{i}_sdist"] = logits.numpy(), sd.numpy()
        out[f"c{i}_idx"], out[f"c{i}_u"], out[f"c{i}_cw"] = idx.numpy().astype(np.int32), u.numpy(), cw.numpy()
    out["num_cases"] = len(cases)
    save("sampler", **out)


def golden_cast_ipe():
    out = {}
    for fam, rays in (("blender", synthetic.blender_rays(12, seed=3)), ("llff", synthetic.llff_rays(12, seed=4))):
        r = to_rays(rays)
        n = 32
        sd = np.sort(np.random.default_rng(5).random((12, n + 1)).astype(np.float32), axis=-1)
        sd[:, 0], sd[:, -1] = 0.0, 1.0
        _, s_to_t = coord.construct_ray_warps(None, r.near, r.far)
        tdist = s_to_t(torch.tensor(sd))
        means, covs = render.cast_rays(tdist, r.origins, r.directions, r.radii, "cone", diag=False)
        basis_t = torch.tensor([[0.0, 0.0, -1.0], [0.0, -1.0, 0.0], [-1.0, 0.0, 0.0]])
        lm, lv = coord.lift_and_diagonalize(means, covs, basis_t)
        feat = coord.integrated_pos_enc(lm, lv, 0, 16)
        mc, cc = render.cast_rays(tdist, r.origins, r.directions, r.radii, "cylinder", diag=False)
        lmc, lvc = coord.lift_and_diagonalize(mc, cc, basis_t)
        for k, v in dict(sdist=sd, tdist=tdist, means=means, covs=covs, lmean=lm, lvar=lv, ipe=feat,
                         cyl_lmean=lmc, cyl_lvar=lvc, origins=r.origins, directions=r.directions,
                         radii=r.radii, near=r.near, far=r.far).items():
            out[f"{fam}_{k}"] = v.numpy() if isinstance(v, torch.Tensor) else v
    # safe_sin straddling 100*pi and big arguments
    x = torch.tensor(np.concatenate([np.linspace(-330, 330, 41), [314.15924, 314.15927, 314.1593, -314.15927, 1e4, -1e4, 2.5e5, -2.5e5]]).astype(np.float32))
    from internal import math as rmath
    out["safe_sin_x"], out["safe_sin_y"] = x.numpy(), rmath.safe_sin(x).numpy()
    save("cast_ipe", **out)


def golden_ide():
    fn = ref_utils.generate_ide_fn(5)
    rng = np.random.default_rng(11)
    d = rng.normal(size=(64, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d[:4] *= np.array([[0.5], [0.9], [1.05], [0.0]])       # non-unit + zero vector
    d[4] = [0, 0, 1]
    d[5] = [0, 0, -1]
    d[6] = [1, 0, 0]
    d = d.astype(np.float32)
    out = {"xyz": d}
    for kap in (0.0, 0.01, 0.1, 0.3, 1.0):
        k = torch.full((64, 1), kap)
        out[f"ide_{kap}"] = fn(torch.tensor(d), k).numpy()
    save("ide", **out)


def golden_mlp():
    out = {}
    pk = dict(seed=0, bias_scale=0.1)
    model, cfg = build_model(param_kw=pk)
    out["param_kw"] = np.array([pk["seed"], pk["bias_scale"], 1.0, 0.0])
    mlp = model.nerf_mlp
    rays = synthetic.blender_rays(6, seed=9)
    r = to_rays(rays)
    n = 8
    sd = torch.linspace(0, 1, n + 1)[None].repeat(6, 1)
    _, s_to_t = coord.construct_ray_warps(None, r.near, r.far)
    tdist = s_to_t(sd)
    means, covs = render.cast_rays(tdist, r.origins, r.directions, r.radii, "cone", diag=False)
    basis_t = torch.tensor([[0.0, 0.0, -1.0], [0.0, -1.0, 0.0], [-1.0, 0.0, 0.0]])
    lm, lv = coord.lift_and_diagonalize(means, covs, basis_t)
    out["lmean"], out["lvar"], out["viewdirs"] = lm.numpy(), lv.numpy(), r.viewdirs.numpy()
    mlp.eval()
    with torch.no_grad():
        res = mlp((means, covs), viewdirs=r.viewdirs)
    for k, v in res.items():
        if v is not None:
            out["eval_" + k] = v.numpy()
    mlp.train()
    res = mlp((means.clone(), covs), viewdirs=r.viewdirs)
    for k, v in res.items():
        if v is not None:
            out["train_" + k] = v.detach().numpy()
    save("mlp", **out)


def golden_mlp_basis():
    """MLP.__call__ on Gaussians with the constructor-default 'icosahedron' / 2 basis (full covariances AND their diagonals as
    input), eval and training mode (density-gradient normals)."""
    pk = dict(seed=8, bias_scale=0.1)
    gin.clear_config()
    gin.parse_config_files_and_bindings([REF_CFG], ["NerfMLP.basis_shape = 'icosahedron'", "NerfMLP.basis_subdivisions = 2"])
    model = models.construct_model(utils.dummy_rays(), configs.Config())
    specs, idx = layout.variant_layout(n_basis=21)
    true = synthetic.make_basis_params(n_basis=21, **pk)[idx]
    sd = model.nerf_mlp.state_dict()
    for sp in specs:
        sd[sp.name + ".weight"].copy_(torch.tensor(true[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim].reshape(sp.out_dim, sp.in_dim)))
        sd[sp.name + ".bias"].copy_(torch.tensor(true[sp.b_off:sp.b_off + sp.out_dim]))
    mlp = model.nerf_mlp
    rays = synthetic.blender_rays(6, seed=19)
    r = to_rays(rays)
    n = 10
    sdist = torch.linspace(0, 1, n + 1)[None].repeat(6, 1)
    _, s_to_t = coord.construct_ray_warps(None, r.near, r.far)
    tdist = s_to_t(sdist)
    means, covs = render.cast_rays(tdist, r.origins, r.directions, r.radii, "cone", diag=False)
    _, covs_diag = render.cast_rays(tdist, r.origins, r.directions, r.radii, "cone", diag=True)
    out = {"param_kw": np.array([pk["seed"], pk["bias_scale"], 1.0, 0.0]), "means": means.numpy(), "covs": covs.numpy(),
           "covs_diag": covs_diag.numpy(), "viewdirs": r.viewdirs.numpy()}
    for tag, cv in (("full", covs), ("diag", torch.diag_embed(covs_diag))):       # a diagonal covariance = the full matrix with zeros
        mlp.eval()
        with torch.no_grad():
            res = mlp((means, cv), viewdirs=r.viewdirs)
        for k, v in res.items():
            if v is not None:
                out[f"{tag}_eval_{k}"] = v.numpy()
        mlp.train()
        res = mlp((means.clone(), cv), viewdirs=r.viewdirs)
        for k, v in res.items():
            if v is not None:
                out[f"{tag}_train_{k}"] = v.detach().numpy()
    save("mlp_basis", **out)


def golden_render():
    out = {}
    rng = np.random.default_rng(13)
    R, N = 10, 48
    density = (rng.random((R, N)) ** 4 * 30).astype(np.float32)
    density[0] = 0.0
    density[1] = 1e4
    tdist = np.sort(rng.random((R, N + 1)).astype(np.float32) * 4 + 2, axis=-1)
    dirs = rng.normal(size=(R, 3)).astype(np.float32)
    rgbs = rng.random((R, N, 3)).astype(np.float32) * 1.4
    dif = rng.random((R, N, 3)).astype(np.float32)
    spc = rng.random((R, N, 3)).astype(np.float32)
    extras = dict(normals=rng.normal(size=(R, N, 3)).astype(np.float32),
                  normals_pred=rng.normal(size=(R, N, 3)).astype(np.float32),
                  roughness=rng.random((R, N, 1)).astype(np.float32),
                  tint=rng.random((R, N, 3)).astype(np.float32))
    far = np.full((R, 1), 6.0, np.float32)
    for k, v in dict(density=density, tdist=tdist, dirs=dirs, rgbs=rgbs, dif=dif, spc=spc, far=far, **extras).items():
        out[k] = v
    for opaque in (False, True):
        w, a, tr = render.compute_alpha_weights(torch.tensor(density), torch.tensor(tdist), torch.tensor(dirs), opaque_background=opaque)
        out[f"weights_opaque{int(opaque)}"] = w.numpy()
    w = torch.tensor(out["weights_opaque0"])
    for mode in ("none", "linear", "norm_linear", "srgb", "norm_srgb"):
        rend = render.volumetric_rendering(torch.tensor(rgbs), torch.tensor(dif), torch.tensor(spc), w, torch.tensor(tdist),
                                           1.0, torch.tensor(far), True, extras={k: torch.tensor(v) for k, v in extras.items()},
                                           srgb_mapping=mode)
        for k, v in rend.items():
            out[f"{mode}_{k}"] = v.numpy()
    save("render", **out)


def run_model(model, cfg, rays, train, gt=None):
    r = to_rays(rays)
    res = {}
    if not train:
        model.eval()
        with torch.no_grad():
            rend, hist = model(r, 1.0, True)
    else:
        model.train()
        model.zero_grad()
        rend, hist = model(r, 1.0, True)
        batch = utils.Batch(rays=r, rgb=gt)
        data_loss, stats = train_utils.compute_data_loss(batch, rend, r, cfg)
        o_loss = train_utils.orientation_loss(r, model, hist, cfg)
        n_loss = train_utils.predicted_normal_loss(model, hist, cfg)
        loss = data_loss + o_loss + n_loss
        loss.backward()
        res["loss_data"], res["loss_orientation"], res["loss_normal"] = data_loss.item(), o_loss.item(), n_loss.item()
        res["loss_total"] = loss.item()
        grads = np.zeros(layout.NUM_PARAMS, np.float32)
        named = dict(model.nerf_mlp.named_parameters())
        for spec in layout.PARAM_SPECS:
            grads[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = named[spec.name + ".weight"].grad.numpy().reshape(-1)
            grads[spec.b_off:spec.b_off + spec.out_dim] = named[spec.name + ".bias"].grad.numpy()
        res["grads"] = grads
    for lvl, (rd, hs) in enumerate(zip(rend, hist)):
        for k, v in rd.items():
            res[f"L{lvl}_r_{k}"] = v.detach().numpy()
        for k, v in hs.items():
            if v is not None:
                res[f"L{lvl}_h_{k}"] = v.detach().numpy()
    return res


def golden_models():
    cases = {
        # name: (gin bindings, param kw, rays, train)
        "model_blender_eval": ([], dict(seed=0, bias_scale=0.05), synthetic.blender_rays(24, seed=1, center_frac=0.4), False),
        "model_blender_sharp_eval": ([], dict(seed=0, bias_scale=0.05, sharpen=20.0), synthetic.blender_rays(24, seed=1, center_frac=0.4), False),
        "model_c1_eval": (["Model.num_levels = 1", "Model.num_nerf_samples = 64"], dict(seed=3, bias_scale=0.05, sharpen=10.0),
                          synthetic.blender_rays(16, seed=2, center_frac=0.4), False),
        "model_llff_linear_eval": (["NerfMLP.srgb_mapping = False", "Config.srgb_mapping_when_rendering = True",
                                    "Config.srgb_mapping_type = 'norm_linear'", "Config.near = 0.", "Config.far = 1."],
                                   dict(seed=4, bias_scale=0.05, sharpen=20.0), synthetic.llff_rays(16, seed=5), False),
        "model_blender_sharp_train": ([], dict(seed=0, bias_scale=0.05, sharpen=20.0), synthetic.blender_rays(16, seed=6, center_frac=0.4), True),
        "model_llff_linear_train": (["NerfMLP.srgb_mapping = False", "Config.srgb_mapping_when_rendering = True",
                                     "Config.srgb_mapping_type = 'norm_linear'", "Config.near = 0.", "Config.far = 1.",
                                     "Model.num_prop_samples = 64", "Model.num_nerf_samples = 96"],
                                    dict(seed=4, bias_scale=0.05, sharpen=20.0), synthetic.llff_rays(12, seed=7), True),
    }
    for name, (bindings, pk, rays, train) in cases.items():
        model, cfg = build_model(bindings, pk)
        gt = synthetic.target_rgb(rays["origins"].shape[0], seed=2)
        res = run_model(model, cfg, rays, train, gt)
        res["bindings"] = np.array(bindings if bindings else [""])
        res["param_kw"] = np.array([pk.get("seed", 0), pk.get("bias_scale", 0.0), pk.get("sharpen", 1.0), pk.get("roughness_bias", 0.0)])
        for k, v in rays.items():
            res["rays_" + k] = v
        res["gt_rgb"] = gt
        # keep fixtures small: grads are big (4.4 MB) -> store a strided subsample + per-tensor norms
        if "grads" in res:
            g = res.pop("grads")
            res["grads_sub"] = g[::97].copy()
            res["grads_tensor_l2"] = np.array([[np.linalg.norm(g[s.w_off:s.w_off + s.out_dim * s.in_dim]),
                                                np.linalg.norm(g[s.b_off:s.b_off + s.out_dim])] for s in layout.PARAM_SPECS])
        save(name, **res)


def golden_camera():
    """camera_utils.pixels_to_rays (camera_utils.py:502-614): pinhole Blender-style camera and an
    LLFF-style camera with the NDC conversion (camera_utils.py:31-97)."""
    from internal import camera_utils
    out = {}
    rng = np.random.default_rng(21)
    for tag, (w, h, focal, ndc) in dict(blender=(800, 800, 1111.111, False), llff=(1008, 756, 815.0, True)).items():
        n = 96
        px = rng.integers(0, w, n).astype(np.int32)
        py = rng.integers(0, h, n).astype(np.int32)
        px[:4] = [0, w - 1, 0, w - 1]
        py[:4] = [0, 0, h - 1, h - 1]
        intr = np.array([[focal, 0, w / 2.0], [0, focal, h / 2.0], [0, 0, 1.0]])
        pixtocam = np.linalg.inv(intr).astype(np.float32)
        c2w = np.zeros((3, 4), np.float32)
        if ndc:
            c2w[:3, :3] = np.eye(3)
            c2w[:3, 3] = [0.11, -0.07, 0.03]
        else:
            c2w[:3, :3] = synthetic._rot(5).astype(np.float32)
            c2w[:3, 3] = (synthetic._rot(5) @ np.array([0.0, 0.0, 4.0])).astype(np.float32)
        res = camera_utils.pixels_to_rays(px, py, pixtocam, c2w, pixtocam_ndc=pixtocam if ndc else None, xnp=np)
        out[tag + "_pix_x"], out[tag + "_pix_y"] = px, py
        out[tag + "_pixtocam"], out[tag + "_camtoworld"] = pixtocam, c2w
        for k, v in zip(("origins", "directions", "viewdirs", "radii", "imageplane"), res):
            out[f"{tag}_{k}"] = np.asarray(v, np.float32)
    save("camera", **out)


GEOMETRY_BINDINGS = [
    # the loss set of configs/llff_refnerf_geometry_losses.gin (sizes reduced; multipliers raised so that every
    # term moves the gradient by a measurable amount)
    "NerfMLP.srgb_mapping = False", "Config.srgb_mapping_when_rendering = True",
    "Config.srgb_mapping_type = 'norm_linear'", "Config.near = 0.", "Config.far = 1.",
    "Model.num_prop_samples = 64", "Model.num_nerf_samples = 64",
    "Config.sample_noise_size = 6", "Config.sample_noise_angles = 3", "Config.sample_angle_range = 5",
    "Config.consistency_warmup_steps = 0.6",
    "Config.acc_threshold_for_consistency_loss = 0.1", "Config.acc_threshold_for_weights_entropy_loss = 0.1",
    "Config.accumulated_weights_loss_mult = 10.0",
    "Config.consistency_diffuse_coarse_loss_mult = 30.0", "Config.consistency_diffuse_loss_mult = 300.0",
    "Config.consistency_diffuse_loss_type = 'var'",
    "Config.consistency_specular_coarse_loss_mult = 30.0", "Config.consistency_specular_loss_mult = 300.0",
    "Config.consistency_specular_loss_type = 'var'",
    "Config.consistency_normal_coarse_loss_mult = 0.003", "Config.consistency_normal_loss_mult = 0.03",
    "Config.consistency_normal_loss_target = 'normals'",
    "Config.consistency_distance_coarse_loss_mult = 0.003", "Config.consistency_distance_loss_mult = 0.03",
    "Config.weights_entropy_coarse_loss_mult = 0.003", "Config.weights_entropy_loss_mult = 0.03",
    "Config.orientation_coarse_loss_mult = 0.01", "Config.orientation_loss_mult = 0.1",
    "Config.predicted_normal_coarse_loss_mult = 3e-3", "Config.predicted_normal_loss_mult = 3e-2",
    "Config.interlevel_loss_mult = 0.0", "Config.distortion_loss_mult = 0.0",
]


def golden_geometry():
    """One training step of NeRFSystem.training_step (nerf_system.py:77-188) with the full regulariser set of
    llff_refnerf_geometry_losses.gin: clean pass, noisy-ray pass, all losses, backward.  Two variants of the
    colour-consistency measure ('var' as shipped, 'mse' + normals_pred target + sRGB MLP colours)."""
    from internal import sample_utils
    variants = {
        "geometry_var": (GEOMETRY_BINDINGS, synthetic.llff_rays(14, seed=11)),
        "geometry_mse_srgb": ([b for b in GEOMETRY_BINDINGS if "srgb_mapping" not in b and "loss_type" not in b
                               and "loss_target" not in b and "Config.near" not in b and "Config.far" not in b
                               and "consistency_diffuse" not in b and "consistency_specular" not in b] + [
            "Config.consistency_diffuse_coarse_loss_mult = 3000.0", "Config.consistency_diffuse_loss_mult = 30000.0",
            "Config.consistency_specular_coarse_loss_mult = 3000.0", "Config.consistency_specular_loss_mult = 30000.0",
            "Config.consistency_diffuse_loss_type = 'mse'", "Config.consistency_specular_loss_type = 'avg_mse'",
            "Config.consistency_normal_loss_target = 'normals_pred'"], synthetic.blender_rays(14, seed=12, center_frac=0.4)),
    }
    for name, (bindings, rays) in variants.items():
        pk = dict(seed=4, bias_scale=0.05, sharpen=20.0)
        model, cfg = build_model(bindings, pk)
        model.train()
        model.zero_grad()
        r = to_rays(rays)
        R = rays["origins"].shape[0]
        gt = synthetic.target_rgb(R, seed=2)
        step = int(0.8 * cfg.max_steps)
        ratio = min(1., step / (cfg.consistency_warmup_steps * cfg.max_steps))
        rend, hist = model(r, 1.0, True)
        torch.manual_seed(5)
        n = cfg.sample_noise_size // cfg.patch_size ** 2
        noisy = sample_utils.sample_noisy_rays(r, rend[-1], cfg.sample_angle_range, n, cfg.sample_noise_angles, ratio)
        rend_n, hist_n = model(noisy, 1.0, True)
        batch = utils.Batch(rays=r, rgb=gt)
        losses = {}
        losses["data"], _ = train_utils.compute_data_loss(batch, rend, r, cfg)
        losses["orientation"] = train_utils.orientation_loss(r, model, hist, cfg)
        losses["predicted_normals"] = train_utils.predicted_normal_loss(model, hist, cfg)
        (losses["diffuse_consistency"], losses["specular_consistency"],
         losses["normals_consistency"]) = train_utils.noisy_consistency_loss(model, rend, rend_n, cfg, ratio)
        losses["acc"] = train_utils.accumulated_weights_loss(rend, cfg)
        losses["distance_consistency"] = train_utils.noisy_distance_consistency_loss(model, r, noisy, rend, rend_n, cfg, ratio)
        losses["weights_entropy"] = train_utils.weights_entropy_loss(model, rend, hist, cfg, ratio)
        total = torch.sum(torch.stack(list(losses.values())))
        total.backward()
        res = {"loss_" + k: float(v) for k, v in losses.items()}
        res["loss_total"] = float(total)
        res["warmup_ratio"] = ratio
        res["global_step"] = step
        g = np.zeros(layout.NUM_PARAMS, np.float32)
        named = dict(model.nerf_mlp.named_parameters())
        for spec in layout.PARAM_SPECS:
            g[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = named[spec.name + ".weight"].grad.numpy().reshape(-1)
            g[spec.b_off:spec.b_off + spec.out_dim] = named[spec.name + ".bias"].grad.numpy()
        res["grads_sub"] = g[::97].copy()
        res["grads_l2"] = float(np.linalg.norm(g))
        res["grads_tensor_l2"] = np.array([[np.linalg.norm(g[s.w_off:s.w_off + s.out_dim * s.in_dim]),
                                            np.linalg.norm(g[s.b_off:s.b_off + s.out_dim])] for s in layout.PARAM_SPECS])
        # a fixed random projection of the full gradient (64 directions): a second, dense fingerprint
        proj_rng = np.random.default_rng(123)
        res["grads_proj"] = np.array([float(np.dot(g.astype(np.float64), proj_rng.standard_normal(g.size))) for _ in range(16)])
        for lvl, (rd, rdn) in enumerate(zip(rend, rend_n)):
            for k in ("rgb", "diffuse", "specular", "distance", "acc", "normals", "normals_pred"):
                res[f"L{lvl}_r_{k}"] = rd[k].detach().numpy()
                res[f"L{lvl}_noisy_r_{k}"] = rdn[k].detach().numpy()
        res["bindings"] = np.array(bindings)
        res["param_kw"] = np.array([pk["seed"], pk["bias_scale"], pk["sharpen"], 0.0])
        for k, v in rays.items():
            res["rays_" + k] = v
        for k in ("origins", "directions", "viewdirs", "radii", "imageplane", "lossmult", "near", "far", "cam_idx"):
            res["noisy_" + k] = getattr(noisy, k).detach().numpy()
        res["gt_rgb"] = gt
        save(name, **res)


SEED_KEYS = ("r_rgb", "r_diffuse", "r_specular", "r_acc", "r_distance", "r_normals", "r_normals_pred", "r_tint",
             "r_roughness", "weights", "density", "roughness", "rgb", "normals_pred", "tint", "diffuse", "specular")


def seed_array(key_index, level, shape):
    """The upstream gradient used for output `SEED_KEYS[key_index]` of level `level` (test and generator share it)."""
    return np.random.default_rng(1000 + 10 * key_index + level).standard_normal(tuple(shape)).astype(np.float32)


def grad_fingerprint(g):
    rng = np.random.default_rng(123)
    return np.concatenate([[np.linalg.norm(g)], g[::997].astype(np.float64),
                           [float(np.dot(g.astype(np.float64), rng.standard_normal(g.size))) for _ in range(8)]])


def golden_seeds():
    """Autograd of the reference for ONE random upstream gradient on ONE output at a time (both levels):
    pins the generic backward (oracle rn_level_backward / refnerf_level_backward) output by output."""
    variants = {
        "seeds_llff_linear": ([b for b in GEOMETRY_BINDINGS if b.startswith(("NerfMLP", "Model", "Config.srgb", "Config.near", "Config.far"))],
                              synthetic.llff_rays(10, seed=13)),
        "seeds_blender_srgb": (["Model.num_prop_samples = 48", "Model.num_nerf_samples = 64"],
                               synthetic.blender_rays(10, seed=14, center_frac=0.4)),
    }
    for name, (bindings, rays) in variants.items():
        pk = dict(seed=4, bias_scale=0.05, sharpen=20.0)
        model, cfg = build_model(bindings, pk)
        model.train()
        r = to_rays(rays)
        rend, hist = model(r, 1.0, True)
        res = {}
        named = dict(model.nerf_mlp.named_parameters())
        for ki, key in enumerate(SEED_KEYS):
            model.zero_grad()
            loss = 0.
            for lvl in range(len(rend)):
                x = rend[lvl][key[2:]] if key.startswith("r_") else hist[lvl][key]
                loss = loss + (x * torch.tensor(seed_array(ki, lvl, x.shape))).sum()
            loss.backward(retain_graph=True)
            g = np.zeros(layout.NUM_PARAMS, np.float32)
            for spec in layout.PARAM_SPECS:
                if named[spec.name + ".weight"].grad is not None:
                    g[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = named[spec.name + ".weight"].grad.numpy().reshape(-1)
                if named[spec.name + ".bias"].grad is not None:
                    g[spec.b_off:spec.b_off + spec.out_dim] = named[spec.name + ".bias"].grad.numpy()
            res["fp_" + key] = grad_fingerprint(g)
            res["loss_" + key] = float(loss)
        res["bindings"] = np.array(bindings)
        res["param_kw"] = np.array([pk["seed"], pk["bias_scale"], pk["sharpen"], 0.0])
        for k, v in rays.items():
            res["rays_" + k] = v
        save(name, **res)


def golden_io():
    """(a) key names / shapes of the Lightning checkpoint the reference writes (`model.` + Model.state_dict(),
    nerf_system.py:22-33) -- names and shapes only, the tensors are regenerated from seeds in the test;
    (b) the reference's image writers (utils.py:169-189) on seeded arrays, decoded back to pixels."""
    import io
    from PIL import Image
    model, cfg = build_model([], dict(seed=0, bias_scale=0.05))
    sd = model.state_dict()
    out = {"ckpt_keys": np.array(["model." + k for k in sd.keys()]),
           "ckpt_shapes": np.array([";".join(map(str, v.shape)) for v in sd.values()])}
    rng = np.random.default_rng(9)
    img = (rng.random((12, 10, 3)) * 1.4 - 0.2).astype(np.float64)
    img[0, 0, 0] = np.nan
    rough = rng.random((12, 10, 1)).astype(np.float64)
    acc = rng.random((12, 10)).astype(np.float64)
    depth = (rng.random((12, 10)) * 5).astype(np.float64)
    depth[1, 1] = np.nan
    out.update(img=img, rough=rough, acc=acc, depth=depth)
    real_open = utils.open_file

    def capture(fn, *a, **k):
        buf = io.BytesIO()
        buf.close = lambda: None
        utils.open_file = lambda pth, mode: buf
        try:
            fn(*a, "x", **k)
        finally:
            utils.open_file = real_open
        buf.seek(0)
        return np.array(Image.open(buf))
    out["png_rgb"] = capture(utils.save_img_u8, torch.tensor(img))
    out["png_rho_masked"] = capture(utils.save_img_u8, torch.tensor(rough), mask=torch.tensor(acc))
    out["tiff_depth"] = capture(utils.save_img_f32, torch.tensor(depth))
    save("io", **out)


def _set_mlp_params(mlp, blob):
    sd = mlp.state_dict()
    for spec in layout.PARAM_SPECS:
        sd[spec.name + ".weight"].copy_(torch.tensor(blob[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim].reshape(spec.out_dim, spec.in_dim)))
        sd[spec.name + ".bias"].copy_(torch.tensor(blob[spec.b_off:spec.b_off + spec.out_dim]))


def _mlp_grads(mlp):
    g = np.zeros(layout.NUM_PARAMS, np.float32)
    named = dict(mlp.named_parameters())
    for spec in layout.PARAM_SPECS:
        if named[spec.name + ".weight"].grad is not None:
            g[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = named[spec.name + ".weight"].grad.numpy().reshape(-1)
        if named[spec.name + ".bias"].grad is not None:
            g[spec.b_off:spec.b_off + spec.out_dim] = named[spec.name + ".bias"].grad.numpy()
    return g


def golden_propmlp():
    """Model.single_mlp = False: a separate PropMLP of the same Ref-NeRF architecture for the proposal level
    (models.py:120-123,236) trained by the interlevel loss (train_utils.py:150-162, stepfun.py:67-89) next to the
    data / orientation / predicted-normal terms.  Gradients of BOTH networks."""
    nerf_b = [ln.strip() for ln in open(REF_CFG) if ln.startswith("NerfMLP.")]
    bindings = ["Model.single_mlp = False", "Config.interlevel_loss_mult = 1.0", "Model.num_prop_samples = 48",
                "Model.num_nerf_samples = 64"] + [b.replace("NerfMLP.", "PropMLP.", 1) for b in nerf_b] + [
                    "PropMLP.density_bias = -3.0"]     # a thin proposal density: the envelope is violated, the loss is > 0
    model, cfg = build_model(bindings, dict(seed=4, bias_scale=0.05, sharpen=20.0))
    assert model.prop_mlp is not model.nerf_mlp
    _set_mlp_params(model.prop_mlp, synthetic.make_params(seed=5, bias_scale=0.05, sharpen=2.0))
    model.train()
    model.zero_grad()
    rays = synthetic.blender_rays(12, seed=15, center_frac=0.4)
    r = to_rays(rays)
    gt = synthetic.target_rgb(12, seed=2)
    rend, hist = model(r, 1.0, False)
    batch = utils.Batch(rays=r, rgb=gt)
    losses = {}
    losses["data"], _ = train_utils.compute_data_loss(batch, rend, r, cfg)
    losses["interlevel"] = train_utils.interlevel_loss(hist, cfg)
    losses["orientation"] = train_utils.orientation_loss(r, model, hist, cfg)
    losses["predicted_normals"] = train_utils.predicted_normal_loss(model, hist, cfg)
    total = torch.sum(torch.stack(list(losses.values())))
    total.backward()
    res = {"loss_" + k: float(v) for k, v in losses.items()}
    res["loss_total"] = float(total)
    res["fp_nerf"] = grad_fingerprint(_mlp_grads(model.nerf_mlp))
    res["fp_prop"] = grad_fingerprint(_mlp_grads(model.prop_mlp))
    # the interlevel term alone (its gradient reaches the proposal network only)
    model.zero_grad()
    rend, hist = model(r, 1.0, False)
    train_utils.interlevel_loss(hist, cfg).backward()
    res["fp_prop_interlevel_only"] = grad_fingerprint(_mlp_grads(model.prop_mlp))
    res["nerf_grad_interlevel_only_l2"] = float(np.linalg.norm(_mlp_grads(model.nerf_mlp)))
    for lvl, rd in enumerate(rend):
        res[f"L{lvl}_r_rgb"] = rd["rgb"].detach().numpy()
        res[f"L{lvl}_h_weights"] = hist[lvl]["weights"].detach().numpy()
        res[f"L{lvl}_h_sdist"] = hist[lvl]["sdist"].detach().numpy()
    res["bindings"] = np.array(bindings)
    res["param_kw"] = np.array([4, 0.05, 20.0, 0.0])
    res["prop_param_kw"] = np.array([5, 0.05, 2.0, 0.0])
    for k, v in rays.items():
        res["rays_" + k] = v
    res["gt_rgb"] = gt
    save("propmlp_interlevel", **res)


def golden_dilation():
    """Model options of the mip-NeRF 360 sampler on the Ref-NeRF architecture: dilation of the proposal step
    function (models.py:167-186, stepfun.py:102-131) and weight annealing (models.py:188-201, train_frac = 0.3)."""
    bindings = ["Model.dilation_bias = 0.0025", "Model.dilation_multiplier = 0.5", "Model.anneal_slope = 10.",
                "Model.num_prop_samples = 64", "Model.num_nerf_samples = 64"]
    pk = dict(seed=0, bias_scale=0.05, sharpen=20.0)
    model, cfg = build_model(bindings, pk)
    model.eval()
    rays = synthetic.blender_rays(16, seed=21, center_frac=0.4)
    with torch.no_grad():
        rend, hist = model(to_rays(rays), 0.3, True)
    res = {"train_frac": 0.3}
    for lvl, (rd, hs) in enumerate(zip(rend, hist)):
        for k in ("rgb", "acc", "distance", "distance_mean"):
            res[f"L{lvl}_r_{k}"] = rd[k].numpy()
        for k in ("sdist", "weights", "density"):
            res[f"L{lvl}_h_{k}"] = hs[k].numpy()
    res["bindings"] = np.array(bindings)
    res["param_kw"] = np.array([pk["seed"], pk["bias_scale"], pk["sharpen"], 0.0])
    for k, v in rays.items():
        res["rays_" + k] = v
    save("model_dilation_anneal_eval", **res)


def analytic_target(rays):
    """Ground-truth colours of a synthetic shiny scene for the rays of synthetic.blender_rays: a unit sphere at the
    origin with a normal-dependent albedo and a Phong highlight, white background (own code; gives the reference
    something with real geometry to fit)."""
    o, d = rays["origins"].astype(np.float64), rays["viewdirs"].astype(np.float64)
    b = (o * d).sum(-1)
    c = (o * o).sum(-1) - 1.0
    disc = b * b - c
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0.0))
    p = o + t[:, None] * d
    n = p / np.maximum(np.linalg.norm(p, axis=-1, keepdims=True), 1e-9)
    light = np.array([0.5, 0.6, 0.62])
    light /= np.linalg.norm(light)
    refl = d - 2.0 * (d * n).sum(-1, keepdims=True) * n
    spec = np.maximum(0.0, refl @ light) ** 24 * 0.8
    col = (0.5 + 0.4 * n) * np.maximum(0.15, n @ light)[:, None] + spec[:, None]
    return np.where(hit[:, None], np.clip(col, 0.0, 1.0), 1.0).astype(np.float32)


TRAINED_BLOB = os.path.join(HERE, "trained_blob.npz")


def golden_trained(steps=400, n_rays=512, n_samples=32, lr=5e-4):
    """"Trained-like" weights: the REFERENCE takes `steps` Adam steps (its own forward, its three Ref-NeRF losses,
    its autograd) on the analytic shiny sphere above, starting from the seeded init.  The resulting blob is stored
    rounded to float16 (trained_blob.npz, 2.2 MB) and that rounded blob DEFINES the fixture weights: the reference
    outputs below are computed after loading the rounded blob back into the reference model."""
    pk = dict(seed=0, bias_scale=0.0)
    model, cfg = build_model([f"Model.num_prop_samples = {n_samples}", f"Model.num_nerf_samples = {n_samples}"], pk)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=lr, eps=1e-6)
    import time
    t0 = time.time()
    hist_loss = []
    for it in range(steps):
        rays = synthetic.blender_rays(n_rays, seed=1000 + it, center_frac=0.85)
        gt = analytic_target(rays)
        r = to_rays(rays)
        opt.zero_grad()
        rend, hist = model(r, 1.0, False)
        batch = utils.Batch(rays=r, rgb=gt)
        data_loss, stats = train_utils.compute_data_loss(batch, rend, r, cfg)
        loss = data_loss + train_utils.orientation_loss(r, model, hist, cfg) + train_utils.predicted_normal_loss(model, hist, cfg)
        loss.backward()
        opt.step()
        hist_loss.append(float(data_loss))
        if it % 20 == 0:
            print(f"step {it}: data loss {float(data_loss):.5f}  ({time.time() - t0:.0f} s)", flush=True)
    blob = np.zeros(layout.NUM_PARAMS, np.float32)
    sd = model.nerf_mlp.state_dict()
    for spec in layout.PARAM_SPECS:
        blob[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = sd[spec.name + ".weight"].numpy().reshape(-1)
        blob[spec.b_off:spec.b_off + spec.out_dim] = sd[spec.name + ".bias"].numpy()
    blob16 = blob.astype(np.float16)
    assert np.isfinite(blob16.astype(np.float32)).all()
    np.savez_compressed(TRAINED_BLOB, blob_f16=blob16, data_loss_curve=np.array(hist_loss, np.float32),
                        recipe=np.array([steps, n_rays, n_samples, lr]))
    print("wrote", TRAINED_BLOB, os.path.getsize(TRAINED_BLOB) // 1024, "KiB; final data loss", hist_loss[-1])
    golden_trained_models()


TRAINED_LONG_BLOB = os.path.join(HERE, "trained_long_blob.npz")


def golden_trained_long(steps=2500, n_rays=384, n_samples=48, lr=1e-3):
    """A harsher "trained-like" weight set (the 400-step blob is gentle on 16-bit operands): the REFERENCE takes
    `steps` Adam steps at twice the learning rate on the analytic shiny sphere; the blob is stored in FLOAT32 (weights off
    every 16-bit grid).  Then its reference outputs: eval at C2's sample counts, one training step."""
    pk = dict(seed=1, bias_scale=0.0)
    model, cfg = build_model([f"Model.num_prop_samples = {n_samples}", f"Model.num_nerf_samples = {n_samples}"], pk)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=lr, eps=1e-6)
    import time
    t0 = time.time()
    hist_loss = []
    for it in range(steps):
        rays = synthetic.blender_rays(n_rays, seed=5000 + it, center_frac=0.85)
        gt = analytic_target(rays)
        r = to_rays(rays)
        opt.zero_grad()
        rend, hist = model(r, 1.0, False)
        batch = utils.Batch(rays=r, rgb=gt)
        data_loss, stats = train_utils.compute_data_loss(batch, rend, r, cfg)
        loss = data_loss + train_utils.orientation_loss(r, model, hist, cfg) + train_utils.predicted_normal_loss(model, hist, cfg)
        loss.backward()
        opt.step()
        hist_loss.append(float(data_loss))
        if it % 50 == 0:
            print(f"step {it}: data loss {float(data_loss):.5f}  ({time.time() - t0:.0f} s)", flush=True)
    blob = np.zeros(layout.NUM_PARAMS, np.float32)
    sd = model.nerf_mlp.state_dict()
    for spec in layout.PARAM_SPECS:
        blob[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = sd[spec.name + ".weight"].numpy().reshape(-1)
        blob[spec.b_off:spec.b_off + spec.out_dim] = sd[spec.name + ".bias"].numpy()
    assert np.isfinite(blob).all()
    np.savez_compressed(TRAINED_LONG_BLOB, blob_f32=blob, data_loss_curve=np.array(hist_loss, np.float32),
                        recipe=np.array([steps, n_rays, n_samples, lr]))
    print("wrote", TRAINED_LONG_BLOB, os.path.getsize(TRAINED_LONG_BLOB) // 1024, "KiB; final data loss", hist_loss[-1],
          "| max |w|", float(np.abs(blob).max()))
    golden_trained_long_models()


def golden_trained_long_models():
    blob = np.load(TRAINED_LONG_BLOB)["blob_f32"]
    cases = {
        "model_trained_long_eval": ([], synthetic.blender_rays(32, seed=41, center_frac=0.8), False),
        "model_trained_long_train": (["Model.num_prop_samples = 64", "Model.num_nerf_samples = 96"],
                                     synthetic.blender_rays(16, seed=42, center_frac=0.8), True),
    }
    for name, (bindings, rays, train) in cases.items():
        model, cfg = build_model_blob(bindings, blob)
        gt = analytic_target(rays)
        res = run_model(model, cfg, rays, train, gt)
        print(name, "density max", float(res["L1_h_density"].max()), "specular max", float(res["L1_h_specular"].max()))
        _finish_model_fixture(name, res, bindings, rays, gt, param_kw=np.array([-2.0, 0.0, 1.0, 0.0]))   # seed -2: trained_long_blob.npz


def analytic_target_ndc(rays):
    """The same kind of ground truth for forward-facing NDC rays (synthetic.llff_rays: origins on z = -1, t in [0, 1] runs to
    z = +1): a shaded sphere inside the NDC cube in front of a smooth two-tone backdrop (own code)."""
    o, d = rays["origins"].astype(np.float64), rays["directions"].astype(np.float64)
    c, rad = np.array([-0.15, 0.1, 0.25]), 0.5
    oc = o - c
    a = (d * d).sum(-1)
    b = (oc * d).sum(-1)
    disc = b * b - a * ((oc * oc).sum(-1) - rad * rad)
    hit = disc > 0
    t = (-b - np.sqrt(np.where(hit, disc, 0.0))) / a
    hit &= (t > 0) & (t < 1)
    p = o + t[:, None] * d
    n = (p - c) / rad
    light = np.array([0.4, 0.7, -0.6])
    light /= np.linalg.norm(light)
    v = d / np.linalg.norm(d, axis=-1, keepdims=True)
    refl = v - 2.0 * (v * n).sum(-1, keepdims=True) * n
    spec = np.maximum(0.0, refl @ light) ** 24 * 0.7
    col = (0.45 + 0.4 * n) * np.maximum(0.2, n @ light)[:, None] + spec[:, None]
    end = o + d                                                      # where the ray leaves the cube (z = +1)
    back = np.stack([0.55 + 0.25 * np.sin(2.0 * end[:, 0]), 0.5 + 0.2 * np.cos(1.5 * end[:, 1]), 0.6 + 0.0 * end[:, 0]], -1)
    return np.where(hit[:, None], np.clip(col, 0.0, 1.0), back).astype(np.float32)


TRAINED_LLFF_BLOB = os.path.join(HERE, "trained_llff_blob.npz")
LLFF_BINDINGS = ["NerfMLP.srgb_mapping = False", "Config.srgb_mapping_when_rendering = True", "Config.srgb_mapping_type = 'norm_linear'",
                 "Config.near = 0.", "Config.far = 1."]


def golden_trained_llff(steps=1200, n_rays=384, n_samples=48, lr=1e-3):
    """Trained-like weights for the forward-facing configurations (C4 / C5: NDC rays, linear colour, norm_linear render map):
    the REFERENCE takes `steps` Adam steps on the analytic NDC scene above; fp32 blob; then its eval outputs at C4's sample
    counts and one training step."""
    pk = dict(seed=2, bias_scale=0.0)
    model, cfg = build_model(LLFF_BINDINGS + [f"Model.num_prop_samples = {n_samples}", f"Model.num_nerf_samples = {n_samples}"], pk)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=lr, eps=1e-6)
    import time
    t0 = time.time()
    hist_loss = []
    for it in range(steps):
        rays = synthetic.llff_rays(n_rays, seed=7000 + it)
        gt = analytic_target_ndc(rays)
        r = to_rays(rays)
        opt.zero_grad()
        rend, hist = model(r, 1.0, False)
        batch = utils.Batch(rays=r, rgb=gt)
        data_loss, stats = train_utils.compute_data_loss(batch, rend, r, cfg)
        loss = data_loss + train_utils.orientation_loss(r, model, hist, cfg) + train_utils.predicted_normal_loss(model, hist, cfg)
        loss.backward()
        opt.step()
        hist_loss.append(float(data_loss))
        if it % 50 == 0:
            print(f"step {it}: data loss {float(data_loss):.5f}  ({time.time() - t0:.0f} s)", flush=True)
    blob = np.zeros(layout.NUM_PARAMS, np.float32)
    sd = model.nerf_mlp.state_dict()
    for spec in layout.PARAM_SPECS:
        blob[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = sd[spec.name + ".weight"].numpy().reshape(-1)
        blob[spec.b_off:spec.b_off + spec.out_dim] = sd[spec.name + ".bias"].numpy()
    assert np.isfinite(blob).all()
    np.savez_compressed(TRAINED_LLFF_BLOB, blob_f32=blob, data_loss_curve=np.array(hist_loss, np.float32),
                        recipe=np.array([steps, n_rays, n_samples, lr]))
    print("wrote", TRAINED_LLFF_BLOB, "final data loss", hist_loss[-1], "| max |w|", float(np.abs(blob).max()))
    cases = {"model_trained_llff_eval": (LLFF_BINDINGS, synthetic.llff_rays(32, seed=51), False),
             "model_trained_llff_train": (LLFF_BINDINGS + ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 96"], synthetic.llff_rays(16, seed=52), True)}
    for name, (bindings, rays, train) in cases.items():
        model, cfg = build_model_blob(bindings, blob)
        gt = analytic_target_ndc(rays)
        res = run_model(model, cfg, rays, train, gt)
        print(name, "density max", float(res["L1_h_density"].max()))
        _finish_model_fixture(name, res, bindings, rays, gt, param_kw=np.array([-3.0, 0.0, 1.0, 0.0]))   # seed -3: trained_llff_blob.npz


def golden_trajectory(steps=20, n_rays=256, n_samples=48, lr=5e-4):
    """The first `steps` optimiser steps of the REFERENCE from the seeded init (its forward, its three Ref-NeRF losses, its
    autograd, torch.optim.Adam) on fixed synthetic batches: the per-step losses and a subsample of the final parameters.
    Checks the whole loop on the other side -- weight re-pack after every step, gradient accumulation into the flat blob,
    optimiser coupling -- not just single steps."""
    pk = dict(seed=3, bias_scale=0.0)
    model, cfg = build_model([f"Model.num_prop_samples = {n_samples}", f"Model.num_nerf_samples = {n_samples}"], pk)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=lr, eps=1e-6)
    out = {"total": [], "data": [], "orientation": [], "normal": []}
    gts = []
    for it in range(steps):
        rays = synthetic.blender_rays(n_rays, seed=9100 + it, center_frac=0.85)
        gt = analytic_target(rays)
        gts.append(gt)
        r = to_rays(rays)
        opt.zero_grad()
        rend, hist = model(r, 1.0, False)
        batch = utils.Batch(rays=r, rgb=gt)
        data_loss, _ = train_utils.compute_data_loss(batch, rend, r, cfg)
        o_loss = train_utils.orientation_loss(r, model, hist, cfg)
        n_loss = train_utils.predicted_normal_loss(model, hist, cfg)
        loss = data_loss + o_loss + n_loss
        loss.backward()
        opt.step()
        for k, v in (("total", loss), ("data", data_loss), ("orientation", o_loss), ("normal", n_loss)):
            out[k].append(float(v))
        print(it, float(loss), flush=True)
    blob = np.zeros(layout.NUM_PARAMS, np.float32)
    sd = model.nerf_mlp.state_dict()
    for spec in layout.PARAM_SPECS:
        blob[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = sd[spec.name + ".weight"].numpy().reshape(-1)
        blob[spec.b_off:spec.b_off + spec.out_dim] = sd[spec.name + ".bias"].numpy()
    init = synthetic.make_params(**pk)
    save("trajectory", recipe=np.array([steps, n_rays, n_samples, lr, 1e-6, pk["seed"]]), final_params_sub=blob[::97].copy(),
         update_sub=(blob - init)[::97].copy(), gt_rgb=np.stack(gts),
         **{"loss_" + k: np.array(v, np.float64) for k, v in out.items()})


def _load_trained_blob():
    return np.load(TRAINED_BLOB)["blob_f16"].astype(np.float32)


def build_model_blob(bindings, blob):
    gin.clear_config()
    gin.parse_config_files_and_bindings([REF_CFG], list(bindings))
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg)
    _set_mlp_params(model.nerf_mlp, blob)
    return model, cfg


def golden_trained_models():
    """Reference outputs on the trained-like weights (trained_blob.npz): eval at C2's sample counts and one
    training step (losses + autograd gradients)."""
    blob = _load_trained_blob()
    cases = {
        "model_trained_eval": ([], synthetic.blender_rays(32, seed=31, center_frac=0.8), False),
        "model_trained_train": (["Model.num_prop_samples = 64", "Model.num_nerf_samples = 96"],
                                synthetic.blender_rays(16, seed=32, center_frac=0.8), True),
    }
    for name, (bindings, rays, train) in cases.items():
        model, cfg = build_model_blob(bindings, blob)
        gt = analytic_target(rays)
        res = run_model(model, cfg, rays, train, gt)
        if train:   # a usable fixture exercises every tensor: no dead branch (e.g. a saturated specular sigmoid)
            g = res["grads"]
            dead = [s.name for s in layout.PARAM_SPECS if not np.any(g[s.w_off:s.w_off + s.out_dim * s.in_dim])]
            assert not dead, f"trained blob has dead tensors: {dead}"
        print(name, "specular max", float(res["L1_h_specular"].max()), "tint range", float(res["L1_h_tint"].min()), float(res["L1_h_tint"].max()))
        _finish_model_fixture(name, res, bindings, rays, gt, param_kw=np.array([-1.0, 0.0, 1.0, 0.0]))


def _finish_model_fixture(name, res, bindings, rays, gt, param_kw):
    res["bindings"] = np.array(bindings if bindings else [""])
    res["param_kw"] = param_kw          # seed -1: weights = trained_blob.npz (float16 -> float32)
    for k, v in rays.items():
        res["rays_" + k] = v
    res["gt_rgb"] = gt
    if "grads" in res:
        g = res.pop("grads")
        res["grads_sub"] = g[::97].copy()
        res["grads_tensor_l2"] = np.array([[np.linalg.norm(g[s.w_off:s.w_off + s.out_dim * s.in_dim]),
                                            np.linalg.norm(g[s.b_off:s.b_off + s.out_dim])] for s in layout.PARAM_SPECS])
    save(name, **res)


def golden_shiny():
    """BASELINE configs[2] (C3): the "shiny" network -- raw_roughness.bias = -6, roughness ~ 1e-3, so the degree-8 /
    degree-16 IDE terms are barely attenuated (ref_utils.py:98-161, models.py:637-665) -- at 192 + 192 samples."""
    b = ["Model.num_prop_samples = 192", "Model.num_nerf_samples = 192"]
    pk = dict(seed=0, bias_scale=0.05, sharpen=20.0, roughness_bias=-6.0)
    cases = {
        "model_shiny_eval": (b, synthetic.blender_rays(16, seed=41, center_frac=0.4), False),
        "model_shiny_train": (b, synthetic.blender_rays(10, seed=42, center_frac=0.4), True),
    }
    for name, (bindings, rays, train) in cases.items():
        model, cfg = build_model(bindings, pk)
        gt = synthetic.target_rgb(rays["origins"].shape[0], seed=2)
        res = run_model(model, cfg, rays, train, gt)
        assert res["L1_h_roughness"].max() < 5e-3, res["L1_h_roughness"].max()
        _finish_model_fixture(name, res, bindings, rays, gt,
                              param_kw=np.array([pk["seed"], pk["bias_scale"], pk["sharpen"], pk["roughness_bias"]]))



VARIANT_FLAGS = ["NerfMLP.net_width_viewdirs = 128", "NerfMLP.use_n_dot_v = False", "NerfMLP.use_specular_tint = False",
                 "NerfMLP.enable_pred_roughness = False"]


def golden_variant_models():
    """The NerfMLP variants this build serves by embedding (layout.variant_layout): reference outputs with the four
    flags of VARIANT_FLAGS at once (eval; one training step with losses and autograd gradients, stored in the VARIANT's
    own flat order), and a training step with disable_density_normals on top (no 'normals' in the history, so no
    predicted-normal loss).  Weights: the embedded elements of a synthetic canonical blob."""
    pk = dict(seed=5, bias_scale=0.05, sharpen=20.0)
    canon = synthetic.make_params(**pk)
    specs, idx = layout.variant_layout(128, False, False, False)
    small = ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 96"]
    cases = {
        "model_variant_eval": (VARIANT_FLAGS, synthetic.blender_rays(16, seed=51, center_frac=0.4), False, True),
        "model_variant_train": (VARIANT_FLAGS + small, synthetic.blender_rays(12, seed=52, center_frac=0.4), True, True),
        "model_variant_nonormals_train": (VARIANT_FLAGS + small + ["NerfMLP.disable_density_normals = True"],
                                          synthetic.blender_rays(12, seed=53, center_frac=0.4), True, False),
    }
    for name, (bindings, rays, train, with_normal_loss) in cases.items():
        gin.clear_config()
        gin.parse_config_files_and_bindings([REF_CFG], list(bindings))
        cfg = configs.Config()
        model = models.construct_model(utils.dummy_rays(), cfg)
        sd = model.nerf_mlp.state_dict()
        assert len(sd) == 2 * len(specs), (len(sd), len(specs))
        true = canon[idx]
        for sp in specs:
            w = true[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim].reshape(sp.out_dim, sp.in_dim)
            assert tuple(sd[sp.name + ".weight"].shape) == w.shape, sp
            sd[sp.name + ".weight"].copy_(torch.tensor(w))
            sd[sp.name + ".bias"].copy_(torch.tensor(true[sp.b_off:sp.b_off + sp.out_dim]))
        r = to_rays(rays)
        gt = synthetic.target_rgb(rays["origins"].shape[0], seed=2)
        res = {}
        if not train:
            model.eval()
            with torch.no_grad():
                rend, hist = model(r, 1.0, True)
        else:
            model.train()
            model.zero_grad()
            rend, hist = model(r, 1.0, True)
            batch = utils.Batch(rays=r, rgb=gt)
            data_loss, _ = train_utils.compute_data_loss(batch, rend, r, cfg)
            o_loss = train_utils.orientation_loss(r, model, hist, cfg)
            loss = data_loss + o_loss
            res["loss_data"], res["loss_orientation"] = data_loss.item(), o_loss.item()
            if with_normal_loss:
                n_loss = train_utils.predicted_normal_loss(model, hist, cfg)
                loss = loss + n_loss
                res["loss_normal"] = n_loss.item()
            loss.backward()
            res["loss_total"] = loss.item()
            named = dict(model.nerf_mlp.named_parameters())
            g = np.zeros(len(idx), np.float32)
            for sp in specs:
                g[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim] = named[sp.name + ".weight"].grad.numpy().reshape(-1)
                g[sp.b_off:sp.b_off + sp.out_dim] = named[sp.name + ".bias"].grad.numpy()
            res["grads_sub"] = g[::61].copy()
            res["grads_tensor_l2"] = np.array([[np.linalg.norm(g[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim]),
                                                np.linalg.norm(g[sp.b_off:sp.b_off + sp.out_dim])] for sp in specs])
        for lvl, (rd, hs) in enumerate(zip(rend, hist)):
            for k, v in rd.items():
                res[f"L{lvl}_r_{k}"] = v.detach().numpy()
            for k, v in hs.items():
                if v is not None:
                    res[f"L{lvl}_h_{k}"] = v.detach().numpy()
        res["history_keys"] = np.array(sorted(k for k, v in hist[-1].items()))
        res["rendering_keys"] = np.array(sorted(rend[-1].keys()))
        res["bindings"] = np.array(bindings)
        res["param_kw"] = np.array([pk["seed"], pk["bias_scale"], pk["sharpen"], 0.0])
        for k, v in rays.items():
            res["rays_" + k] = v
        res["gt_rgb"] = gt
        save(name, **res)
        print(name, "history keys", list(res["history_keys"]), "rendering keys", list(res["rendering_keys"]))


def golden_basis_models():
    """NerfMLP.basis_shape / basis_subdivisions other than 'octahedron' / 1 (internal/models.py:384-385, 482-484;
    geopoly.generate_basis): the reference's constructor default 'icosahedron' / 2 (21 directions, 672 IPE features) in eval
    mode and for one training step, and 'icosahedron' / 1 (6 directions) in eval mode; otherwise blender_refnerf.gin.
    Also tests/golden/geopoly.npz: the reference's basis matrices themselves.  Weights: synthetic.make_basis_params
    (extended canonical blob), gradients in the module's own flat order."""
    from internal import geopoly
    save("geopoly", **{f"{shape}_{v}_{int(rs)}": geopoly.generate_basis(shape, v, rs)
                       for shape in ("octahedron", "icosahedron") for v in (1, 2, 3) for rs in (True, False)})
    pk = dict(seed=7, bias_scale=0.05, sharpen=20.0)
    small = ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 96"]
    ico2 = ["NerfMLP.basis_shape = 'icosahedron'", "NerfMLP.basis_subdivisions = 2"]
    ico1 = ["NerfMLP.basis_shape = 'icosahedron'", "NerfMLP.basis_subdivisions = 1"]
    # narrower networks / fewer IPE degrees (dead units and zero columns of the canonical network: layout.variant_layout)
    narrow = ["NerfMLP.net_width = 128", "NerfMLP.bottleneck_width = 64", "NerfMLP.min_deg_point = 1", "NerfMLP.max_deg_point = 12",
              "NerfMLP.net_width_viewdirs = 192"]
    narrow_kw = dict(net_width=128, bottleneck_width=64, min_deg_point=1, max_deg_point=12, net_width_viewdirs=192)
    cases = {"model_ico_eval": (ico2, dict(n_basis=21), synthetic.blender_rays(16, seed=71, center_frac=0.4), False),
             "model_ico_train": (ico2 + small, dict(n_basis=21), synthetic.blender_rays(12, seed=72, center_frac=0.4), True),
             "model_ico1_eval": (ico1, dict(n_basis=6), synthetic.blender_rays(16, seed=73, center_frac=0.4), False),
             "model_narrow_eval": (narrow, narrow_kw, synthetic.blender_rays(16, seed=74, center_frac=0.4), False),
             "model_narrow_train": (narrow + small, narrow_kw, synthetic.blender_rays(12, seed=75, center_frac=0.4), True),
             "model_narrow_ico1_train": (narrow + ico1 + small, dict(narrow_kw, n_basis=6), synthetic.blender_rays(12, seed=76, center_frac=0.4), True),
             # shallower trunks: identity layers of the canonical network (layout.identity_fill)
             "model_shallow_eval": (["NerfMLP.net_depth = 3", "NerfMLP.net_depth_viewdirs = 7"], dict(net_depth=3, net_depth_viewdirs=7),
                                    synthetic.blender_rays(16, seed=77, center_frac=0.4), False),
             "model_shallow_train": (["NerfMLP.net_depth = 6", "NerfMLP.net_depth_viewdirs = 2", "NerfMLP.net_width = 192"] + small,
                                     dict(net_depth=6, net_depth_viewdirs=2, net_width=192), synthetic.blender_rays(12, seed=78, center_frac=0.4), True)}
    only = os.environ.get("GOLDEN_ONLY")
    for name, (bindings, lkw, rays, train) in cases.items():
        if only and only not in name:
            continue
        n_basis = lkw.get("n_basis", 3)
        specs, idx = layout.variant_layout(**lkw)
        canon = synthetic.make_basis_params(n_basis=n_basis, **pk) if n_basis != 3 else synthetic.make_params(**pk)
        gin.clear_config()
        gin.parse_config_files_and_bindings([REF_CFG], list(bindings))
        cfg = configs.Config()
        model = models.construct_model(utils.dummy_rays(), cfg)
        sd = model.nerf_mlp.state_dict()
        assert len(sd) == 2 * len(specs), (len(sd), len(specs))
        true = canon[idx]
        for sp in specs:
            w = true[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim].reshape(sp.out_dim, sp.in_dim)
            assert tuple(sd[sp.name + ".weight"].shape) == w.shape, (sp, sd[sp.name + ".weight"].shape)
            sd[sp.name + ".weight"].copy_(torch.tensor(w))
            sd[sp.name + ".bias"].copy_(torch.tensor(true[sp.b_off:sp.b_off + sp.out_dim]))
        r = to_rays(rays)
        gt = synthetic.target_rgb(rays["origins"].shape[0], seed=2)
        res = {}
        if not train:
            model.eval()
            with torch.no_grad():
                rend, hist = model(r, 1.0, True)
        else:
            model.train()
            model.zero_grad()
            rend, hist = model(r, 1.0, True)
            batch = utils.Batch(rays=r, rgb=gt)
            data_loss, _ = train_utils.compute_data_loss(batch, rend, r, cfg)
            o_loss = train_utils.orientation_loss(r, model, hist, cfg)
            n_loss = train_utils.predicted_normal_loss(model, hist, cfg)
            loss = data_loss + o_loss + n_loss
            res["loss_data"], res["loss_orientation"], res["loss_normal"] = data_loss.item(), o_loss.item(), n_loss.item()
            loss.backward()
            res["loss_total"] = loss.item()
            named = dict(model.nerf_mlp.named_parameters())
            g = np.zeros(len(idx), np.float32)
            for sp in specs:
                g[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim] = named[sp.name + ".weight"].grad.numpy().reshape(-1)
                g[sp.b_off:sp.b_off + sp.out_dim] = named[sp.name + ".bias"].grad.numpy()
            res["grads_sub"] = g[::61].copy()
            res["grads_tensor_l2"] = np.array([[np.linalg.norm(g[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim]),
                                                np.linalg.norm(g[sp.b_off:sp.b_off + sp.out_dim])] for sp in specs])
        for lvl, (rd, hs) in enumerate(zip(rend, hist)):
            for k, v in rd.items():
                res[f"L{lvl}_r_{k}"] = v.detach().numpy()
            for k, v in hs.items():
                if v is not None:
                    res[f"L{lvl}_h_{k}"] = v.detach().numpy()
        res["bindings"] = np.array(bindings)
        res["n_basis"] = n_basis
        res["layout_kw"] = np.array(json.dumps(lkw))
        res["basis"] = model.nerf_mlp.pos_basis_t.numpy().T.copy()
        res["param_kw"] = np.array([pk["seed"], pk["bias_scale"], pk["sharpen"], 0.0])
        for k, v in rays.items():
            res["rays_" + k] = v
        res["gt_rgb"] = gt
        save(name, **res)


def golden_posenc_models():
    """`NerfMLP.use_directional_enc = False`: coord.pos_enc of the reflected direction instead of the IDE (models.py:487-492),
    otherwise the Ref-NeRF config -- eval and one training step, gradients in the variant's own flat order."""
    pk = dict(seed=6, bias_scale=0.05, sharpen=20.0)
    canon = synthetic.make_params(**pk)
    flags = ["NerfMLP.use_directional_enc = False"]
    specs, idx = layout.variant_layout(use_directional_enc=False, deg_view=5)
    small = ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 96"]
    cases = {"model_posenc_eval": (flags, synthetic.blender_rays(16, seed=61, center_frac=0.4), False),
             "model_posenc_train": (flags + small, synthetic.blender_rays(12, seed=62, center_frac=0.4), True)}
    for name, (bindings, rays, train) in cases.items():
        gin.clear_config()
        gin.parse_config_files_and_bindings([REF_CFG], list(bindings))
        cfg = configs.Config()
        model = models.construct_model(utils.dummy_rays(), cfg)
        sd = model.nerf_mlp.state_dict()
        true = canon[idx]
        for sp in specs:
            w = true[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim].reshape(sp.out_dim, sp.in_dim)
            assert tuple(sd[sp.name + ".weight"].shape) == w.shape, (sp, tuple(sd[sp.name + ".weight"].shape))
            sd[sp.name + ".weight"].copy_(torch.tensor(w))
            sd[sp.name + ".bias"].copy_(torch.tensor(true[sp.b_off:sp.b_off + sp.out_dim]))
        gt = synthetic.target_rgb(rays["origins"].shape[0], seed=2)
        res = run_model(model, cfg, rays, False, gt) if not train else None
        if train:
            r = to_rays(rays)
            model.train()
            model.zero_grad()
            rend, hist = model(r, 1.0, True)
            batch = utils.Batch(rays=r, rgb=gt)
            data_loss, _ = train_utils.compute_data_loss(batch, rend, r, cfg)
            o_loss = train_utils.orientation_loss(r, model, hist, cfg)
            n_loss = train_utils.predicted_normal_loss(model, hist, cfg)
            loss = data_loss + o_loss + n_loss
            loss.backward()
            res = {"loss_data": data_loss.item(), "loss_orientation": o_loss.item(), "loss_normal": n_loss.item(), "loss_total": loss.item()}
            named = dict(model.nerf_mlp.named_parameters())
            g = np.zeros(len(idx), np.float32)
            for sp in specs:
                if named[sp.name + ".weight"].grad is None:      # raw_roughness: pos_enc ignores the roughness (no gradient)
                    assert sp.name == "raw_roughness", sp.name
                    continue
                g[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim] = named[sp.name + ".weight"].grad.numpy().reshape(-1)
                g[sp.b_off:sp.b_off + sp.out_dim] = named[sp.name + ".bias"].grad.numpy()
            res["grads_sub"] = g[::61].copy()
            res["grads_tensor_l2"] = np.array([[np.linalg.norm(g[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim]),
                                                np.linalg.norm(g[sp.b_off:sp.b_off + sp.out_dim])] for sp in specs])
            for lvl, (rd, hs) in enumerate(zip(rend, hist)):
                for k, v in rd.items():
                    res[f"L{lvl}_r_{k}"] = v.detach().numpy()
                for k, v in hs.items():
                    if v is not None:
                        res[f"L{lvl}_h_{k}"] = v.detach().numpy()
        res["bindings"] = np.array(bindings)
        res["param_kw"] = np.array([pk["seed"], pk["bias_scale"], pk["sharpen"], 0.0])
        for k, v in rays.items():
            res["rays_" + k] = v
        res["gt_rgb"] = gt
        save(name, **res)


RAYDIST_CASES = {
    # name: (raydist_fn attribute set on the reference Model, disable_integration, rays, train)
    "model_raydist_reciprocal_eval": ("reciprocal", False, "blender", False),
    "model_raydist_log_eval": ("log", False, "blender", False),
    "model_raydist_piecewise_eval": ("piecewise", False, "llff", False),
    "model_nointegration_eval": (None, True, "blender", False),
    "model_raydist_nointegration_train": ("reciprocal", True, "blender", True),
}


def golden_raydist_models():
    """Model.raydist_fn (coord.construct_ray_warps, coord.py:63-99: None / 'piecewise' / torch.reciprocal / log ...) and
    Model.disable_integration (models.py:228-231): the sample positions along the ray and the zero-covariance IPE -- eval
    per function, and one training step with both switched on."""
    fns = {None: None, "piecewise": "piecewise", "reciprocal": torch.reciprocal, "log": torch.log}
    small = ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 96"]
    for name, (fn, noint, family, train) in RAYDIST_CASES.items():
        bindings = list(small) if train else []
        if family == "llff":
            bindings += ["Config.near = 0.", "Config.far = 1."]
        pk = dict(seed=7, bias_scale=0.05, sharpen=20.0)
        model, cfg = build_model(bindings, pk)
        model.raydist_fn = fns[fn]                      # read at call time (models.py:147)
        model.disable_integration = noint
        rays = (synthetic.llff_rays(12, seed=71) if family == "llff" else
                synthetic.blender_rays(12 if train else 16, seed=72, center_frac=0.4))
        gt = synthetic.target_rgb(rays["origins"].shape[0], seed=2)
        res = run_model(model, cfg, rays, train, gt)
        res["raydist_fn"] = np.array(fn if fn else "")
        res["disable_integration"] = np.array(int(noint))
        _finish_model_fixture(name, res, bindings, rays, gt, param_kw=np.array([pk["seed"], pk["bias_scale"], pk["sharpen"], 0.0]))


def golden_variants():
    """Which of the reference's shipped configs construct and run at all.

    The mip-NeRF configs switch `use_diffuse_color` off; `Model.__call__` reads `ray_results['diffuse']` unconditionally
    (internal/models.py:272), so `construct_model` -- which runs the model once -- raises for them.  Recorded as data: the
    exception each config ends in (or 'ok'), plus, for a Ref-NeRF config with single flags flipped, whether the reference
    still runs.  tests/test_host_cpu.py::test_variant_gate_matches_reference_status replays it against this build's gate.
    """
    import json
    import traceback
    cfg_dir = os.path.join(_ref_harness.REFERENCE_ROOT, "configs")
    status = {}
    for name in sorted(os.listdir(cfg_dir)):
        gin.clear_config()
        try:
            gin.parse_config_files_and_bindings([os.path.join(cfg_dir, name)], [])
            cfg = configs.Config()
            m = models.construct_model(utils.dummy_rays(), cfg)
            m.eval()
            m(utils.dummy_rays(), 1.0, True)
            status[name] = "ok"
        except Exception as e:   # noqa: BLE001
            tb = traceback.extract_tb(e.__traceback__)[-1]
            status[name] = f"{type(e).__name__}: {e} (internal/{os.path.basename(tb.filename)}:{tb.lineno})"
    flags = {}
    for flag in ["NerfMLP.use_diffuse_color = False", "NerfMLP.use_directional_enc = False", "NerfMLP.use_reflections = False",
                 "NerfMLP.enable_pred_roughness = False", "NerfMLP.use_specular_tint = False", "NerfMLP.use_n_dot_v = False",
                 "NerfMLP.net_width_viewdirs = 128", "NerfMLP.basis_shape = 'icosahedron'", "NerfMLP.disable_density_normals = True",
                 "NerfMLP.enable_pred_normals = False"]:
        gin.clear_config()
        try:
            gin.parse_config_files_and_bindings([REF_CFG], [flag])
            cfg = configs.Config()
            m = models.construct_model(utils.dummy_rays(), cfg)
            m.eval()
            m(utils.dummy_rays(), 1.0, True)
            flags[flag] = "ok"
        except Exception as e:   # noqa: BLE001
            tb = traceback.extract_tb(e.__traceback__)[-1]
            flags[flag] = f"{type(e).__name__}: {e} (internal/{os.path.basename(tb.filename)}:{tb.lineno})"
    path = os.path.join(HERE, "variants_status.json")
    with open(path, "w") as f:
        json.dump({"configs": status, "refnerf_with_flag": flags}, f, indent=1, sort_keys=True)
    print("wrote", path)
    for k, v in {**status, **flags}.items():
        print(f"  {k}: {v}")


def golden_normal_metrics():
    """`Config.compute_normal_metrics` / `compute_disp_metrics`: the statistics branch of train_utils.compute_data_loss
    (internal/train_utils.py:62-84) on seeded two-level renderings -- with normals (training mode) and without (NaN)."""
    rng = np.random.default_rng(9)
    R = 40
    gin.clear_config()
    gin.parse_config_files_and_bindings([REF_CFG], ["Config.compute_normal_metrics = True", "Config.compute_disp_metrics = True"])
    cfg = configs.Config()
    rays = synthetic.blender_rays(R, seed=3, center_frac=0.4)
    r = to_rays(rays)
    out = {}
    for tag, with_normals in (("train", True), ("eval", False)):
        rend = []
        for lvl in range(2):
            d = dict(rgb=torch.tensor(rng.random((R, 3)).astype(np.float32)), acc=torch.tensor(rng.random(R).astype(np.float32)),
                     distance_mean=torch.tensor((2 + 4 * rng.random(R)).astype(np.float32)))
            if with_normals:
                d["normals"] = torch.tensor(rng.standard_normal((R, 3)).astype(np.float32))
            rend.append(d)
        batch = utils.Batch(rays=r, rgb=rng.random((R, 3)).astype(np.float32), disps=torch.tensor(rng.random(R).astype(np.float32)),
                            normals=torch.tensor(rng.standard_normal((R, 3)).astype(np.float32)), alphas=torch.tensor(rng.random(R).astype(np.float32)))
        loss, stats = train_utils.compute_data_loss(batch, rend, r, cfg)
        out[tag + "_loss"] = float(loss)
        for k, v in stats.items():
            out[f"{tag}_stat_{k}"] = v.numpy()
        for lvl, d in enumerate(rend):
            for k, v in d.items():
                out[f"{tag}_L{lvl}_{k}"] = v.numpy()
        out[tag + "_gt_rgb"] = batch.rgb
        out[tag + "_disps"], out[tag + "_normals"], out[tag + "_alphas"] = batch.disps.numpy(), batch.normals.numpy(), batch.alphas.numpy()
    for k, v in rays.items():
        out["rays_" + k] = v
    save("normal_metrics", **out)


SPECDENS_FLAGS = ["NerfMLP.enable_pred_specular_density = True", "Config.render_with_specular_density = True"]


def specdens_head(seed=77):
    """the extra head's parameters (not part of the canonical 46-tensor blob): seeded, stored in the fixture"""
    rng = np.random.default_rng(seed)
    return (rng.uniform(-1, 1, (1, layout.WIDTH)) / np.sqrt(layout.WIDTH) * 12.0).astype(np.float32), np.array([0.3], np.float32)


def golden_specdens_models():
    """`NerfMLP.enable_pred_specular_density` + `Config.render_with_specular_density` (internal/models.py:250-258,502-503,583-584,
    624-625,745-746): the extra `specular_density` entry of ray_history, the unchanged renderings, and -- training -- that the head
    receives no gradient (nothing reads `specular_weights`)."""
    small = ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 64"]
    cases = {
        "model_specdens_eval": (SPECDENS_FLAGS, synthetic.blender_rays(16, seed=61, center_frac=0.4), False),
        "model_specdens_train": (SPECDENS_FLAGS + small, synthetic.blender_rays(12, seed=62, center_frac=0.4), True),
    }
    pk = dict(seed=0, bias_scale=0.05, sharpen=20.0)
    w, b = specdens_head()
    for name, (bindings, rays, train) in cases.items():
        model, cfg = build_model(bindings, pk)
        sd = model.nerf_mlp.state_dict()
        assert tuple(sd["raw_specular_density.weight"].shape) == w.shape
        sd["raw_specular_density.weight"].copy_(torch.tensor(w))
        sd["raw_specular_density.bias"].copy_(torch.tensor(b))
        gt = synthetic.target_rgb(rays["origins"].shape[0], seed=2)
        res = run_model(model, cfg, rays, train, gt)
        assert "L1_h_specular_density" in res
        res["hist_keys"] = np.array(list(model(to_rays(rays), 1.0, True)[1][0].keys()))
        res["state_dict_keys"] = np.array(list(model.nerf_mlp.state_dict().keys()))
        if train:
            res["specdens_grad_is_none"] = np.array([model.nerf_mlp.raw_specular_density.weight.grad is None,
                                                     model.nerf_mlp.raw_specular_density.bias.grad is None])
        res["specdens_w"], res["specdens_b"] = w, b
        res["bindings"] = np.array(bindings)
        res["param_kw"] = np.array([pk.get("seed", 0), pk.get("bias_scale", 0.0), pk.get("sharpen", 1.0), pk.get("roughness_bias", 0.0)])
        for k, v in rays.items():
            res["rays_" + k] = v
        res["gt_rgb"] = gt
        if "grads" in res:
            g = res.pop("grads")
            res["grads_sub"] = g[::97].copy()
            res["grads_tensor_l2"] = np.array([[np.linalg.norm(g[s.w_off:s.w_off + s.out_dim * s.in_dim]),
                                                np.linalg.norm(g[s.b_off:s.b_off + s.out_dim])] for s in layout.PARAM_SPECS])
        save(name, **res)


if __name__ == "__main__":
    which = sys.argv[1:] or ["sampler", "cast_ipe", "ide", "mlp", "render", "models", "camera", "geometry", "seeds", "io", "propmlp", "dilation", "shiny", "trained_models", "variants", "variant_models", "posenc_models", "raydist_models", "basis_models", "mlp_basis", "specdens_models", "normal_metrics"]
    for w in which:
        globals()["golden_" + w]()
