"""GPU: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors captured from the reference.  Run with `pytest -m gpu` on an MI355X."""
import numpy as np
import pytest
import torch

from helpers import (EVAL_CASES, HIST_KEYS, REND_KEYS, TRAIN_CASES, cfg_from_bindings, load_golden,
                     params_from_golden, rays_from_golden)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import refnerf_pl_amd  # noqa: F401
    from refnerf_pl_amd import _hip
    _hip.require_device()          # fails loudly: no fallback
    return _hip


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


DEV = "cuda:0"


def dev_rays(rays):
    out = {}
    for k, v in rays.items():
        t = torch.tensor(np.asarray(v, np.float32), device=DEV)
        out[k] = t.reshape(-1) if k in ("radii", "near", "far") else t
    return out


def run_hip_model(hip, P, rays, kw, lv, precision=0):
    # training-mode levels read the f32 image (it carries the bf16 chain ops), the f16x2 chains of the built-in basis their own
    packed = hip.pack_weights(torch.tensor(P, device=DEV), precision=hip.level_image(precision, bool(kw.get("training")), int(kw.get("ipe_groups", 0))))
    r = dev_rays(rays)
    R = rays["origins"].shape[0]
    sdist = torch.tensor([[0.0, 1.0]], device=DEV).repeat(R, 1)
    weights = torch.ones((R, 1), device=DEV)
    nl = lv.get("num_levels", 2)
    outs = []
    for L in range(nl):
        n = lv.get("num_prop_samples", 128) if L < nl - 1 else lv.get("num_nerf_samples", 128)
        cfg = hip.default_cfg(n_samples=n, n_in=weights.shape[1], precision=precision, **kw)
        res = hip.level_forward(packed, cfg, r, sdist, weights)
        sdist, weights = res["sdist"], res["weights"]
        outs.append({k: v.cpu().numpy() for k, v in res.items()})
    return outs


# ---------------------------------------------------------------- stages
def test_sampler_bit_exact_vs_oracle_and_golden(hip, O):
    """Identical (t, logits) in -> CDF bin indices AND sdist bit-identical to the
    oracle (shared exp, sequential CDF), indices identical to the reference."""
    g = load_golden("sampler")
    for i in range(int(g["num_cases"])):
        t, lg, n = g[f"c{i}_t"], g[f"c{i}_logits"], int(g[f"c{i}_n"])
        sd, idx = hip.sample_intervals(torch.tensor(t, device=DEV), torch.tensor(lg, device=DEV), n)
        sd_o, idx_o = O.sample_intervals(t, lg, n)
        assert np.array_equal(idx.cpu().numpy(), idx_o), f"case {i}: bin indices differ from oracle"
        assert np.array_equal(idx.cpu().numpy(), g[f"c{i}_idx"]), f"case {i}: bin indices differ from reference"
        assert np.array_equal(sd.cpu().numpy(), sd_o), f"case {i}: sdist not bit-equal to oracle"


def test_sampler_large_random_bit_exact(hip, O):
    rng = np.random.default_rng(3)
    R, M, N = 512, 128, 128
    t = np.sort(rng.random((R, M + 1)).astype(np.float32), axis=-1)
    t[:, 0], t[:, -1] = 0, 1
    w = (rng.random((R, M)) ** 6).astype(np.float32)
    lg = O.resample_logits(t, w)
    sd, idx = hip.sample_intervals(torch.tensor(t, device=DEV), torch.tensor(lg, device=DEV), N)
    sd_o, idx_o = O.sample_intervals(t, lg, N)
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    assert np.array_equal(sd.cpu().numpy(), sd_o)
    s = sd.cpu().numpy()
    assert np.all(np.diff(s, axis=-1) >= 0) and s.min() >= 0 and s.max() <= 1


def test_sampler_errors(hip):
    t = torch.zeros((1, 2), device=DEV)
    with pytest.raises(ValueError, match="num_samples must be > 1"):      # stepfun.py:234-235
        hip.sample_intervals(t, torch.zeros((1, 1), device=DEV), 1)


@pytest.mark.parametrize("fam", ["blender", "llff"])
def test_ipe_stage(hip, fam):
    g = load_golden("cast_ipe")
    f = hip.integrated_pos_enc(torch.tensor(g[fam + "_lmean"], device=DEV), torch.tensor(g[fam + "_lvar"], device=DEV))
    np.testing.assert_allclose(f.cpu().numpy(), g[fam + "_ipe"], rtol=0, atol=1e-6)


def test_ide_stage_gated_by_fp64(hip, O):
    """|hip - fp64| <= |reference - fp64| + 1e-5 at every roughness."""
    g = load_golden("ide")
    xyz = g["xyz"]
    for kap in (0.0, 0.01, 0.1, 0.3, 1.0):
        mine = hip.integrated_dir_enc(torch.tensor(xyz, device=DEV), torch.full((xyz.shape[0], 1), kap, device=DEV)).cpu().numpy()
        truth = O.ide(xyz, kap, "f64")
        ref = g[f"ide_{kap}"]
        assert np.abs(mine - truth).max() <= np.abs(ref - truth).max() + 1e-5
        assert np.abs(mine - truth).max() < 5e-6
        np.testing.assert_allclose(mine, O.ide(xyz, kap, "stable"), rtol=0, atol=1e-6)


# ---------------------------------------------------------------- end to end
STRICT_INDEX_CASES = ("model_blender_eval", "model_blender_sharp_eval", "model_c1_eval", "model_llff_linear_eval",
                      "model_blender_sharp_train", "model_llff_linear_train")
WIDE_RANGE_CASES = ("model_trained_eval", "model_trained_train", "model_shiny_eval", "model_shiny_train")


def hist_tol(name, L, k):
    """(rtol, atol) of the per-sample comparison HIP f32 mode <-> oracle / reference.  Random-init fixtures: round-off.
    Shiny (un-attenuated degree-16 IDE terms in a K = 457 GEMM) and trained-like weights (pre-activations of 20-30):
    summation-order noise reaches 1e-5 per sample at level 0; at level 1 the sample POSITIONS already differ by an ulp
    (sdist ~1e-7) and the trained network turns that into <= 5e-3 per sample -- the rendered
    quantities, compared separately, stay at 1e-5."""
    tol = 2e-4 if k == "normals_pred" else (1e-4 if k == "density" else 5e-6)
    if name not in WIDE_RANGE_CASES:
        return 0.0, tol
    if k in ("sdist", "weights"):
        return 0.0, (1e-5 if L == 0 else 5e-5)
    if L == 0:
        return 2e-5, max(tol, 1e-5)
    return 2e-4, max(tol, 5e-3 if name.startswith("model_trained") else 5e-5)
def test_render_rays_all_render_time_modes(hip):
    """refnerf_render_rays = compute_alpha_weights + volumetric_rendering (render.py:132-254): the five render-time
    srgb_mapping modes (:186-216), extras and float64 percentiles against the REFERENCE's outputs on the same
    per-sample inputs (tests/golden/render.npz) -- the same checker the oracle passes on CPU."""
    from test_oracle_golden import check_render_modes
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=DEV)

    def fn(mode, g):
        cfg = hip.default_cfg(render_srgb_mode=mode)
        res = hip.render_rays(cfg, t(g["density"]), t(g["tdist"]), t(g["dirs"]), t(g["far"]), t(g["rgbs"]), t(g["dif"]),
                              t(g["spc"]), t(g["normals"]), t(g["normals_pred"]), t(g["roughness"]), t(g["tint"]))
        return {k: v.cpu().numpy() for k, v in res.items()}
    worst = check_render_modes(fn)
    print("render_rays vs reference, worst abs error per output over the five modes:", worst)
    # opaque background (render.py:139-143) through the same entry
    g = load_golden("render")
    res = hip.render_rays(hip.default_cfg(opaque_background=1), t(g["density"]), t(g["tdist"]), t(g["dirs"]), t(g["far"]))
    np.testing.assert_allclose(res["weights"].cpu().numpy(), g["weights_opaque1"], rtol=0, atol=2e-7)
    with pytest.raises(ValueError):
        hip.render_rays(hip.default_cfg(), torch.zeros((2, 2000), device=DEV), torch.zeros((2, 2001), device=DEV),
                        torch.zeros((2, 3), device=DEV), torch.zeros(2, device=DEV))


@pytest.mark.parametrize("name", EVAL_CASES)
def test_level_forward_vs_oracle_and_reference(hip, O, name):
    g = load_golden(name)
    P = params_from_golden(g)
    rays = rays_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    ref = O.model_forward(P, rays, **lv, **kw)
    outs = run_hip_model(hip, P, rays, kw, lv)
    ok = np.ones(rays["origins"].shape[0], bool)
    for L, (res, orc) in enumerate(zip(outs, ref)):
        if L == 0:
            assert np.array_equal(res["sdist"], orc["sdist"])
        # sample indices: identical to the oracle on the random-init fixtures (fp32 mode); on the shiny / trained-like
        # ones a level-1 quantile within an ulp of a CDF knot may take the neighbouring bin: bounded,
        # and the per-sample comparison continues on the rays whose indices agree
        same = res["bin_idx"] == orc["bin_idx"]
        assert np.mean(same) == 1.0 if name in STRICT_INDEX_CASES else np.mean(same) >= 0.999, (L, np.mean(same))
        ok &= same.all(-1)
        assert ok.mean() >= 0.9
        for k in HIST_KEYS:
            rtol, tol = hist_tol(name, L, k)
            a, b, c = res[k][ok], orc[k].reshape(res[k].shape)[ok], g[f"L{L}_h_{k}"].reshape(res[k].shape)[ok]
            np.testing.assert_allclose(a, b, rtol=rtol, atol=tol, err_msg=f"L{L} {k}")
            np.testing.assert_allclose(a, c, rtol=rtol, atol=max(tol, 2e-5), err_msg=f"golden L{L} {k}")
        for k in REND_KEYS:
            rtol_r = 2e-5 if name in WIDE_RANGE_CASES else 1e-5
            if k == "distance_mean":     # exp(sum(w log t) / acc): round-off scales with 1 / acc (rays that miss everything)
                rtol_r = rtol_r + 1e-6 / np.maximum(orc["r_acc"][ok], 1e-6)
            for want in (orc["r_" + k][ok], g[f"L{L}_r_{k}"].reshape(res["r_" + k].shape)[ok]):
                diff = np.abs(res["r_" + k][ok] - want)
                lim = rtol_r if np.ndim(rtol_r) == 0 or diff.ndim == 1 else rtol_r[:, None]
                assert np.all(diff <= lim), (L, k, float(diff.max()))
        # headline bar: RGB L-inf <= 1e-4 vs the reference CPU path (ALL rays, whatever their indices did)
        err = np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max()
        print(f"{name} f32 L{L}: RGB L-inf vs reference {err:.2e}, identical bin indices {np.mean(same):.5f}")
        assert err <= 1e-4
        pc = np.stack([g[f"L{L}_r_distance_percentile_5"], g[f"L{L}_r_distance_median"], g[f"L{L}_r_distance_percentile_95"]], -1)
        assert res["r_percentiles"].dtype == np.float64
        perr = np.abs(res["r_percentiles"][ok] - pc[ok])
        if name in WIDE_RANGE_CASES:   # float64 interpolation inside a near-empty CDF bin: bulk tight, worst case bounded
            assert np.mean(perr <= 2e-5) >= 0.95 and perr.max() <= 5e-4, (np.mean(perr <= 2e-5), perr.max())
        else:
            assert perr.max() <= 2e-5


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_training_forward_density_normals(hip, O, name):
    """Training-mode level forward: density-gradient normals (models.py:603-609)
    through the transposed-weight VJP; everything else as in eval."""
    g = load_golden(name)
    P = params_from_golden(g)
    rays = rays_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    ref = O.model_forward(P, rays, training=1, **lv, **kw)
    outs = run_hip_model(hip, P, rays, dict(kw, training=1), lv)
    ok = np.ones(rays["origins"].shape[0], bool)
    for L, (res, orc) in enumerate(zip(outs, ref)):
        same = res["bin_idx"] == orc["bin_idx"]
        assert np.mean(same) == 1.0 if name in STRICT_INDEX_CASES else np.mean(same) >= 0.999, (L, np.mean(same))
        ok &= same.all(-1)
        assert ok.mean() >= 0.9
        for k in HIST_KEYS:
            rtol, tol = hist_tol(name, L, k)
            np.testing.assert_allclose(res[k][ok], orc[k].reshape(res[k].shape)[ok], rtol=rtol, atol=tol, err_msg=f"L{L} {k}")
        assert np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max() <= 1e-4
        for refn, bulk, tail in ((orc["normals"], 2e-5, 0.99), (g[f"L{L}_h_normals"], 1e-4, 0.97)):
            err = np.abs(res["normals"] - refn.reshape(res["normals"].shape)).max(-1)[ok]
            # ill-conditioned where the density gradient is tiny: bulk tight, tail loose
            assert np.median(err) < bulk and np.mean(err < 1e-3) > tail, (L, np.median(err), np.mean(err < 1e-3))
        np.testing.assert_allclose(res["r_normals"][ok], g[f"L{L}_r_normals"][ok], rtol=0, atol=2e-3)
        np.testing.assert_allclose(res["r_normals"][ok], orc["r_normals"][ok], rtol=0, atol=5e-4)


def _hip_train_step(hip, P, rays, gt_rgb, lossmult, kw, lv, mults):
    """Training forward + torch losses on the level outputs + HIP backward."""
    packed = hip.pack_weights(torch.tensor(P, device=DEV), precision=0)
    r = dev_rays(rays)
    R = rays["origins"].shape[0]
    sdist = torch.tensor([[0.0, 1.0]], device=DEV).repeat(R, 1)
    weights = torch.ones((R, 1), device=DEV)
    gt = torch.tensor(np.asarray(gt_rgb, np.float32)[..., :3].reshape(R, 3), device=DEV)
    lm = torch.tensor(np.asarray(lossmult, np.float32).reshape(R, 1), device=DEV).expand(R, 3)
    grads = torch.zeros(hip.NUM_PARAMS, device=DEV)
    nl = lv.get("num_levels", 2)
    losses = {"data": 0.0, "orientation": 0.0, "normal": 0.0}
    for L in range(nl):
        fine = L == nl - 1
        n = lv.get("num_nerf_samples", 128) if fine else lv.get("num_prop_samples", 128)
        cfg = hip.default_cfg(n_samples=n, n_in=weights.shape[1], precision=0, training=1, **kw)
        res = hip.level_forward(packed, cfg, r, sdist, weights, save_activations=True)
        rgb = res["r_rgb"].clone().requires_grad_(True)
        w = res["weights"].clone().requires_grad_(True)
        npred = res["normals_pred"].clone().requires_grad_(True)
        dm, om, nm = (mults[0][fine], mults[1][fine], mults[2][fine])
        l_data = dm * (lm * (rgb - gt) ** 2).sum() / lm.sum()                      # train_utils.py:33-88
        ndv = (npred * (-r["viewdirs"])[:, None, :]).sum(-1)
        l_or = om * (w * torch.clamp(ndv, max=0.0) ** 2).sum(-1).mean()            # :165-183
        l_nm = nm * (w * (1.0 - (res["normals"] * npred).sum(-1))).sum(-1).mean()  # :186-204
        (l_data + l_or + l_nm).backward()
        hip.level_backward(packed, cfg, r, res, rgb.grad, w.grad, npred.grad, grads)
        losses["data"] += float(l_data.detach()); losses["orientation"] += float(l_or.detach()); losses["normal"] += float(l_nm.detach())
        sdist, weights = res["sdist"], res["weights"]
    torch.cuda.synchronize()
    return losses, grads.cpu().numpy()


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_training_step_gradients(hip, O, name):
    """HIP backward (recompute + transposed chains + split-K weight-gradient
    GEMM) against the oracle's backward and the reference's autograd gradients."""
    from refnerf_pl_amd import layout
    g = load_golden(name)
    P = params_from_golden(g)
    rays = rays_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    mults = ((0.1, 1.0), (0.01, 0.1), (3e-5, 3e-4))
    losses, grads = _hip_train_step(hip, P, rays, g["gt_rgb"], rays["lossmult"], kw, lv, mults)
    o_losses, o_grads, _ = O.model_train(P, rays, g["gt_rgb"], **lv, **kw)
    for k in ("data", "orientation", "normal"):
        assert losses[k] == pytest.approx(o_losses[k], rel=2e-4), k
    assert losses["data"] == pytest.approx(float(g["loss_data"]), rel=1e-5)
    wide = name.startswith("model_trained")   # level-1 sample positions differ by an ulp, amplified ~5e3 x by this network
    rel = np.linalg.norm(grads - o_grads) / np.linalg.norm(o_grads)
    print(f"{name}: gradient rel-L2 vs oracle {rel:.2e}")
    assert rel < (1e-3 if wide else 2e-4), rel      # fp32 summation order over the sample axis differs (split-K MFMA vs sequential)
    ref = g["grads_sub"]
    assert np.linalg.norm(grads[::97] - ref) / np.linalg.norm(ref) < (1e-3 if wide else 2e-4)
    norms = g["grads_tensor_l2"]
    for i, s in enumerate(layout.PARAM_SPECS):       # all 46 tensors receive their gradient
        nw = s.out_dim * s.in_dim
        trel = 3e-2 if wide else 2e-3
        assert np.linalg.norm(grads[s.w_off:s.w_off + nw]) == pytest.approx(norms[i, 0], rel=trel), s.name
        assert np.linalg.norm(grads[s.b_off:s.b_off + s.out_dim]) == pytest.approx(norms[i, 1], rel=trel), s.name
        ow, ob = o_grads[s.w_off:s.w_off + nw], o_grads[s.b_off:s.b_off + s.out_dim]
        # per tensor: round-off level (~6e-7) except where a pre-activation within 1 ulp of 0 takes the
        # other side of the ReLU than in the oracle's summation order (isolated flips, <= ~2e-3 of a tensor)
        # (trained-like fixture: the directional tensors carry gradients of norm ~1e-6 of the whole -- a nearly saturated
        # specular branch -- so their own norm is no yardstick: bound them against the whole gradient as well)
        tb, floor = (5e-2, 1e-4 * np.linalg.norm(o_grads)) if wide else (5e-3, 1e-12)
        assert np.linalg.norm(grads[s.w_off:s.w_off + nw] - ow) <= tb * np.linalg.norm(ow) + floor, s.name
        assert np.linalg.norm(grads[s.b_off:s.b_off + s.out_dim] - ob) <= tb * np.linalg.norm(ob) + floor, s.name


def test_model_api_matches_reference_contract(hip):
    """Model.__call__ through the host mirror: keys, shapes, dtypes and values."""
    from refnerf_pl_amd import configs, models, utils
    g = load_golden("model_blender_sharp_eval")
    configs.clear_config()
    import os
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")], [])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).eval()
    model.nerf_mlp.load_flat_params(params_from_golden(g))
    rays = utils.rays_from_dict(rays_from_golden(g), DEV)
    with torch.no_grad():
        renderings, history = model(rays, 1.0, True)
    assert len(renderings) == len(history) == 2
    R = 24
    expect_r = {"rgb": (R, 3), "diffuse": (R, 3), "specular": (R, 3), "distance": (R, 1), "acc": (R,),
                "normals_pred": (R, 3), "tint": (R, 3), "roughness": (R, 1), "distance_mean": (R,),
                "distance_percentile_5": (R,), "distance_median": (R,), "distance_percentile_95": (R,),
                "ray_sdist": (16, 129), "ray_weights": (16, 128), "ray_rgbs": (16, 128, 3)}
    for L in range(2):
        assert list(renderings[L].keys()) == list(expect_r.keys())
        for k, shp in expect_r.items():
            assert tuple(renderings[L][k].shape) == shp, k
            want = torch.float64 if "percentile" in k or "median" in k else torch.float32
            assert renderings[L][k].dtype == want
            np.testing.assert_allclose(renderings[L][k].cpu().numpy(), g[f"L{L}_r_{k}"], rtol=0, atol=2e-5, err_msg=k)
        assert history[L]["normals"] is None
        for k in ("density", "rgb", "normals_pred", "grad_pred", "tint", "diffuse", "specular", "roughness", "sdist", "weights"):
            assert tuple(history[L][k].shape) == g[f"L{L}_h_{k}"].shape, k
    # render_image on a tiny 6x4 "image" reuses the same rays
    img = utils.rays_from_dict({k: v.reshape(6, 4, -1) for k, v in rays_from_golden(g).items()}, DEV)
    cfg.render_chunk_size = 10
    with torch.no_grad():
        rendering = models.render_image(lambda r: model(r, 1.0, True), img, cfg, verbose=False, device=torch.device(DEV))
    assert rendering["rgb"].shape == (6, 4, 3)
    np.testing.assert_allclose(rendering["rgb"].reshape(-1, 3).cpu().numpy(), g["L1_r_rgb"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("R,nprop,nfine,extra", [(5, 64, 96, {}), (3, 33, 40, {"opaque_background": 1}),
                                                 (9, 128, 64, {"srgb_mapping": 0, "render_srgb_mode": 2})])
def test_training_step_ragged_shapes(hip, O, R, nprop, nfine, extra):
    """Backward at ragged sizes (R not a multiple of the workgroup tile, N not a
    multiple of 32, partially filled passes), opaque background and the
    render-time norm_linear colour map, against the oracle."""
    from refnerf_pl_amd import synthetic
    P = synthetic.make_params(seed=3, bias_scale=0.05, sharpen=8.0)
    rays = synthetic.blender_rays(R, seed=11, center_frac=0.5)
    rays["lossmult"] = (0.5 + synthetic.hash_uniform(5, 9, R)).astype(np.float32).reshape(R, 1)
    gt = synthetic.target_rgb(R, seed=4)
    kw = dict(extra)
    if "render_srgb_mode" in kw:
        kw["render_srgb_mode"] = int(kw["render_srgb_mode"])
    lv = dict(num_levels=2, num_prop_samples=nprop, num_nerf_samples=nfine)
    mults = ((0.1, 1.0), (0.01, 0.1), (3e-5, 3e-4))
    losses, grads = _hip_train_step(hip, P, rays, gt, rays["lossmult"], kw, lv, mults)
    o_losses, o_grads, _ = O.model_train(P, rays, gt, **lv, **kw)
    for k in ("data", "orientation", "normal"):
        assert losses[k] == pytest.approx(o_losses[k], rel=5e-4, abs=1e-9), k
    rel = np.linalg.norm(grads - o_grads) / np.linalg.norm(o_grads)
    assert np.isfinite(grads).all() and rel < 5e-4, rel


def test_training_step_n256_vs_oracle(hip, O):
    """C5 shape per sample count (256 samples per level): two-pass rays in the backward / training forward."""
    from refnerf_pl_amd import synthetic
    R = 3
    P = synthetic.make_params(seed=1, bias_scale=0.05, sharpen=12.0)
    rays = synthetic.llff_rays(R, seed=4)
    rays["lossmult"] = np.ones((R, 1), np.float32)
    gt = synthetic.target_rgb(R, seed=8)
    kw = dict(srgb_mapping=0, render_srgb_mode=2)              # llff_refnerf_geometry_losses.gin colour handling
    lv = dict(num_levels=2, num_prop_samples=256, num_nerf_samples=256)
    mults = ((0.1, 1.0), (0.01, 0.1), (3e-5, 3e-4))
    losses, grads = _hip_train_step(hip, P, rays, gt, rays["lossmult"], kw, lv, mults)
    o_losses, o_grads, _ = O.model_train(P, rays, gt, **lv, **kw)
    for k in ("data", "orientation", "normal"):
        assert losses[k] == pytest.approx(o_losses[k], rel=5e-4, abs=1e-9), k
    assert np.linalg.norm(grads - o_grads) / np.linalg.norm(o_grads) < 5e-4


def test_full_size_training_step_properties(hip):
    """4096 rays x 128 samples x 2 levels through Model + autograd: finite, every tensor receives a
    gradient, and the step is bit-reproducible (split-K partials are reduced in a fixed order)."""
    import os
    from refnerf_pl_amd import configs, layout, models, synthetic, train_utils, utils
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")], [])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).train()
    model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
    R = 4096
    rays = utils.rays_from_dict(synthetic.blender_rays(R, seed=1, center_frac=0.5), DEV)
    batch = utils.Batch(rays=rays, rgb=synthetic.target_rgb(R, seed=7))

    def step():
        model.zero_grad(set_to_none=True)
        rend, hist = model(rays, 1.0, False)
        total, terms, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
        total.backward()
        return float(total.detach()), torch.cat([p.grad.reshape(-1) for p in model.nerf_mlp.ordered_parameters()])
    l1, g1 = step()
    l2, g2 = step()
    assert np.isfinite(l1) and l1 == l2
    assert torch.isfinite(g1).all() and torch.equal(g1, g2)
    for p in model.nerf_mlp.ordered_parameters():
        assert float(p.grad.abs().max()) > 0
    # acc / distance are differentiable outputs too
    model.zero_grad(set_to_none=True)
    rend, _ = model(rays, 1.0, False)
    (rend[1]["acc"].mean() + rend[1]["distance"].mean()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.nerf_mlp.ordered_parameters())


def test_backward_is_linear_in_the_upstream_gradients(hip):
    """Size-independent property at a larger batch: the parameter gradient is linear in
    (dL/d rgb, dL/d weights, dL/d n_pred), accumulates across calls, and a zero seed gives zero."""
    from refnerf_pl_amd import synthetic
    R, N = 320, 128
    P = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0)
    packed = hip.pack_weights(torch.tensor(P, device=DEV), precision=0)
    r = dev_rays(synthetic.blender_rays(R, seed=2, center_frac=0.5))
    sd = torch.tensor([[0.0, 1.0]], device=DEV).repeat(R, 1)
    w = torch.ones((R, 1), device=DEV)
    cfg = hip.default_cfg(n_samples=N, n_in=1, precision=0, training=1)
    res = hip.level_forward(packed, cfg, r, sd, w, save_activations=True)
    g = torch.Generator(device="cpu").manual_seed(1)
    ga = [torch.randn((R, 3), generator=g).to(DEV) * 1e-3, torch.randn((R, N), generator=g).to(DEV) * 1e-3,
          torch.randn((R, N, 3), generator=g).to(DEV) * 1e-3]
    gb = [torch.randn((R, 3), generator=g).to(DEV) * 1e-3, None, None]

    def run(seeds, acc=None):
        out = torch.zeros(hip.NUM_PARAMS, device=DEV) if acc is None else acc
        hip.level_backward(packed, cfg, r, res, seeds[0], seeds[1], seeds[2], out)
        return out
    A, B = run(ga), run(gb)
    AB = run([ga[0] * 2.0 + gb[0] * -3.0, ga[1] * 2.0, ga[2] * 2.0])
    ref = 2.0 * A - 3.0 * B
    assert float((AB - ref).norm() / ref.norm()) < 2e-5
    acc = run(gb, acc=A.clone())                                   # accumulation into an existing blob
    assert float((acc - (A + B)).norm() / (A + B).norm()) < 1e-6
    Z = run([torch.zeros((R, 3), device=DEV), None, None])
    assert float(Z.abs().max()) == 0.0
    A2 = run(ga)                                                   # bit-reproducible (no atomics)
    assert torch.equal(A, A2)
    # acc = sum_i w_i and distance = sum_i w_i t_mid,i: their seeds equal a weights seed
    g_acc = torch.randn((R,), generator=g).to(DEV) * 1e-3
    g_dist = torch.randn((R,), generator=g).to(DEV) * 1e-3
    t = r["near"][:, None] * (1 - res["sdist"]) + r["far"][:, None] * res["sdist"]      # coord.py:96-98, fn = None
    tmid = 0.5 * (t[:, 1:] + t[:, :-1])
    zero = torch.zeros((R, 3), device=DEV)
    via_w = run([zero, g_acc[:, None] + g_dist[:, None] * tmid, None])
    out = torch.zeros(hip.NUM_PARAMS, device=DEV)
    hip.level_backward(packed, cfg, r, res, zero, None, None, out, g_r_acc=g_acc, g_r_distance=g_dist)
    assert float((out - via_w).norm() / via_w.norm()) < 1e-5


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("covform", ["full", "diag"])
def test_mlp_call_stage_entry(hip, mode, covform):
    """MLP.__call__(gaussians=(means, covs), viewdirs) -- the reference's per-sample entry
    (models.py:533-750) -- against the vectors captured from the reference MLP."""
    import os
    from refnerf_pl_amd import configs, models, synthetic, utils
    g = load_golden("mlp")
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")], [])
    model = models.construct_model(utils.dummy_rays(), configs.Config()).to(DEV)
    pk = g["param_kw"]
    model.nerf_mlp.load_flat_params(synthetic.make_params(seed=int(pk[0]), bias_scale=float(pk[1]), sharpen=float(pk[2]),
                                                          roughness_bias=float(pk[3])))
    mlp = model.nerf_mlp.train() if mode == "train" else model.nerf_mlp.eval()
    lm, lv = g["lmean"], g["lvar"]
    means = np.stack([-lm[..., 2], -lm[..., 1], -lm[..., 0]], -1)           # un-lift: basis = [[0,0,-1],[0,-1,0],[-1,0,0]]
    if covform == "full":
        covs = np.zeros(lm.shape[:-1] + (3, 3), np.float32)
        covs[..., 0, 0], covs[..., 1, 1], covs[..., 2, 2] = lv[..., 2], lv[..., 1], lv[..., 0]
        covs[..., 0, 1] = covs[..., 1, 0] = 0.37 * lv[..., 0]              # off-diagonals do not reach the octahedron/1 lift
    else:
        covs = np.stack([lv[..., 2], lv[..., 1], lv[..., 0]], -1)
    with torch.no_grad():
        res = mlp((torch.tensor(means), torch.tensor(covs)), viewdirs=torch.tensor(g["viewdirs"]))
    assert list(res.keys()) == ["density", "rgb", "normals", "normals_pred", "grad_pred", "tint", "diffuse", "specular", "roughness"]
    for k, v in res.items():
        if k == "normals" and mode == "eval":
            assert v is None
            continue
        want = g[f"{mode}_{k}"]
        assert tuple(v.shape) == want.shape, k
        tol = 5e-6 if k in ("normals", "normals_pred") else 2e-6
        np.testing.assert_allclose(v.cpu().numpy(), want, rtol=0, atol=tol, err_msg=k)
    with pytest.raises(ValueError):
        mlp((torch.tensor(means), torch.tensor(covs)), viewdirs=None)


@pytest.mark.parametrize("tag", ["blender", "llff"])
def test_device_ray_generation(hip, tag):
    """camera_utils.pixels_to_rays on the device against the reference's output (pinhole + NDC),
    per-pixel cameras, and cast_pinhole_rays against the numpy generator used for the fixtures."""
    from refnerf_pl_amd import camera_utils, synthetic
    g = load_golden("camera")
    ndc = g[tag + "_pixtocam"] if tag == "llff" else None
    res = camera_utils.pixels_to_rays(g[tag + "_pix_x"], g[tag + "_pix_y"], g[tag + "_pixtocam"], g[tag + "_camtoworld"],
                                      pixtocam_ndc=ndc, device=torch.device(DEV))
    keys = ("origins", "directions", "viewdirs", "radii", "imageplane")
    for k, v in zip(keys, res):
        want = g[f"{tag}_{k}"]
        assert tuple(v.shape) == want.shape, k
        np.testing.assert_allclose(v.cpu().numpy(), want, rtol=2e-6, atol=2e-7, err_msg=k)
    n = g[tag + "_pix_x"].shape[0]
    per = camera_utils.pixels_to_rays(g[tag + "_pix_x"], g[tag + "_pix_y"], np.tile(g[tag + "_pixtocam"], (n, 1, 1)),
                                      np.tile(g[tag + "_camtoworld"], (n, 1, 1)), pixtocam_ndc=ndc, device=torch.device(DEV))
    for a, b in zip(res, per):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        camera_utils.pixels_to_rays(g[tag + "_pix_x"], g[tag + "_pix_y"], g[tag + "_pixtocam"], g[tag + "_camtoworld"],
                                    camtype=camera_utils.ProjectionType.FISHEYE, device=torch.device(DEV))
    if tag == "blender":
        rays = camera_utils.cast_pinhole_rays(g["blender_camtoworld"], 40, 56, 77.0, 2.0, 6.0, device=torch.device(DEV))
        yy, xx = np.meshgrid(np.arange(40), np.arange(56), indexing="ij")
        o, d, v, r, ip = synthetic._pixels_to_rays(xx.reshape(-1), yy.reshape(-1), 77.0, 56, 40, g["blender_camtoworld"].astype(np.float64))
        assert tuple(rays.origins.shape) == (40, 56, 3) and tuple(rays.near.shape) == (40, 56, 1)
        np.testing.assert_allclose(rays.directions.reshape(-1, 3).cpu().numpy(), d, rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(rays.radii.reshape(-1, 1).cpu().numpy(), r, rtol=2e-6, atol=1e-9)
        assert float(rays.far.min()) == 6.0


@pytest.mark.parametrize("precision", [0, 1, 3])        # REFNERF_PREC_F32, _BF16, _F16X2 (the mode of record)
def test_c_abi_without_torch(hip, precision, tmp_path):
    """examples/c_abi_demo.cpp -- a C++ host that uses only include/refnerf_hip.h (hipMalloc'd buffers,
    no PyTorch) -- renders the same view as the Python host mirror, to the last few ulps."""
    import os
    import subprocess
    import __graft_entry__ as ge
    from refnerf_pl_amd import camera_utils, configs, models, synthetic, utils
    exe = ge.build_c_demo()
    W, H, focal = 24, 16, 30.0
    blob = synthetic.make_params(seed=2, bias_scale=0.05, sharpen=20.0)
    c2w, _ = synthetic.blender_camera(seed=3)
    blob.astype(np.float32).tofile(tmp_path / "w.f32")
    c2w.astype(np.float32).tofile(tmp_path / "c2w.f32")
    out = subprocess.run([exe, str(tmp_path / "w.f32"), str(tmp_path / "c2w.f32"), str(W), str(H), str(focal), str(precision),
                          str(tmp_path / "rgb.f32")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    rgb_c = np.fromfile(tmp_path / "rgb.f32", np.float32).reshape(H, W, 3)
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                            ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 64",
                                             f"Config.hip_precision = '{ {0: 'f32', 1: 'bf16', 3: 'f16x2'}[precision] }'"])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).eval()
    model.nerf_mlp.load_flat_params(blob)
    rays = camera_utils.cast_pinhole_rays(c2w.astype(np.float32), H, W, focal, 2.0, 6.0, device=torch.device(DEV))
    with torch.no_grad():
        rend, _ = model(rays.reshape(H * W, -1), 1.0, False)
    rgb_py = rend[1]["rgb"].reshape(H, W, 3).cpu().numpy()
    np.testing.assert_allclose(rgb_c, rgb_py, rtol=0, atol=2e-6)
    assert "mean rgb" in out.stdout


def test_hip_graph_replay(hip):
    """graphs.GraphedForward: the two level launches of an eval step captured in a HIP graph and replayed
    on new rays give exactly the eager results."""
    import os
    from refnerf_pl_amd import configs, graphs, models, synthetic, utils
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                            ["Config.hip_precision = 'bf16'"])
    model = models.construct_model(utils.dummy_rays(), configs.Config()).to(DEV).eval()
    model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
    a = utils.rays_from_dict(synthetic.blender_rays(512, seed=1, center_frac=0.5), DEV)
    b = utils.rays_from_dict(synthetic.blender_rays(512, seed=2, center_frac=0.5), DEV)
    g = graphs.GraphedForward(model, a)
    with torch.no_grad():
        want_b = model(b, 1.0, True)
        want_a = model(a, 1.0, True)
    for rays, want in ((b, want_b), (a, want_a)):
        rend, hist = g(rays)
        torch.cuda.synchronize()
        for L in range(2):
            for k in ("rgb", "acc", "distance_median"):
                assert torch.equal(rend[L][k], want[0][L][k]), k
            assert torch.equal(hist[L]["weights"], want[1][L]["weights"])
    with pytest.raises(ValueError):
        g(utils.rays_from_dict(synthetic.blender_rays(64, seed=1), DEV))
    model.train()
    with pytest.raises(ValueError):
        graphs.GraphedForward(model, a)


def test_model_training_step_autograd(hip):
    """Model.__call__ in training mode + the reference-shaped losses + loss.backward():
    the 46 nn.Parameters receive the reference's gradients (golden autograd vectors)."""
    import os
    from refnerf_pl_amd import configs, layout, models, train_utils, utils
    g = load_golden("model_blender_sharp_train")
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")], [])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).train()
    model.nerf_mlp.load_flat_params(params_from_golden(g))
    rd = rays_from_golden(g)
    rays = utils.rays_from_dict(rd, DEV)
    batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
    renderings, history = model(rays, 1.0, False)
    assert history[0]["normals"] is not None and "normals" not in renderings[0]     # compute_extras=False
    total, terms, stats = train_utils.compute_losses(model, batch, rays, renderings, history, cfg)
    assert float(terms["data"].detach()) == pytest.approx(float(g["loss_data"]), rel=1e-5)
    assert float(terms["orientation"].detach()) == pytest.approx(float(g["loss_orientation"]), rel=2e-4)
    assert float(terms["predicted_normals"].detach()) == pytest.approx(float(g["loss_normal"]), rel=2e-4)
    assert float(total.detach()) == pytest.approx(float(g["loss_total"]), rel=1e-5)
    assert stats["mses"].shape == (2,)
    total.backward()
    flat = torch.zeros(layout.NUM_PARAMS)
    for spec, lin in model.nerf_mlp._named_linears():
        assert lin.weight.grad is not None and lin.bias.grad is not None, spec.name
        flat[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = lin.weight.grad.reshape(-1).cpu()
        flat[spec.b_off:spec.b_off + spec.out_dim] = lin.bias.grad.cpu()
    grads = flat.numpy()
    ref = g["grads_sub"]
    assert np.linalg.norm(grads[::97] - ref) / np.linalg.norm(ref) < 2e-4
    norms = g["grads_tensor_l2"]
    for i, s in enumerate(layout.PARAM_SPECS):
        nw = s.out_dim * s.in_dim
        assert np.linalg.norm(grads[s.w_off:s.w_off + nw]) == pytest.approx(norms[i, 0], rel=2e-3), s.name
    # an optimiser step invalidates the packed weight image; the next forward repacks and changes
    before = renderings[1]["rgb"].detach().clone()
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    opt.step()
    r2, _ = model(rays, 1.0, False)
    assert (r2[1]["rgb"].detach() - before).abs().max() > 0
    # eval mode under no_grad still takes the inference kernel and yields no normals
    model.eval()
    with torch.no_grad():
        _, h3 = model(rays, 1.0, False)
    assert h3[0]["normals"] is None


@pytest.mark.parametrize("name", ["model_variant_eval", "model_variant_train", "model_variant_nonormals_train",
                                  "model_posenc_eval", "model_posenc_train"])
@pytest.mark.parametrize("flat,chains", [(False, "f32"), (True, "f32"), (False, "f16x2")])
def test_nerfmlp_variants_vs_reference(hip, name, flat, chains):
    """the NerfMLP variants the reference runs and this build serves by embedding (net_width_viewdirs = 128,
    no n.v input, no tint head, no roughness head; disable_density_normals) -- Model.__call__ with the reference's gin
    bindings against the reference's own outputs, dict keys, losses and autograd gradients (tests/golden/model_variant_*)."""
    import os
    from helpers import HIST_KEYS, REND_KEYS, VARIANT_HIST_KEYS, VARIANT_REND_KEYS, posenc_params, variant_params
    from refnerf_pl_amd import configs, layout, models, train_utils, utils
    g = load_golden(name)
    train = name.endswith("train")
    if (flat or chains != "f32") and not train:
        pytest.skip("flat gradients / chain modes: training only")
    posenc = "posenc" in name      # `use_directional_enc = False` (coord.pos_enc of the reflected direction), all heads present
    _, true_blob, idx = posenc_params(g) if posenc else variant_params(g)
    hist_keys, rend_keys = (HIST_KEYS, REND_KEYS) if posenc else (VARIANT_HIST_KEYS, VARIANT_REND_KEYS)
    want_specs = layout.variant_layout(use_directional_enc=False)[0] if posenc else layout.variant_layout(128, False, False, False)[0]
    bindings = [str(b) for b in g["bindings"]]
    if "loss_normal" not in g.files:
        bindings += ["Config.predicted_normal_loss_mult = 0.", "Config.predicted_normal_coarse_loss_mult = 0."]
    if flat:
        bindings += ["Config.hip_flat_grads = True"]
    bindings += [f"Config.hip_train_precision = '{chains}'", f"Config.hip_bwd_precision = '{chains}'"]   # f16x2: split-f16 chains, same bars
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")], bindings)
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV)
    mlp = model.nerf_mlp
    assert sorted(n for n, _ in mlp.named_parameters()) == sorted(x for s in want_specs for x in (s.name + ".weight", s.name + ".bias"))
    assert [tuple(p.shape) for p in mlp.ordered_parameters()[::2]] == [(s.out_dim, s.in_dim) for s in want_specs]
    mlp.load_flat_params(true_blob)
    rays = utils.rays_from_dict(rays_from_golden(g), DEV)
    model.train(train)
    with torch.set_grad_enabled(train):
        renderings, history = model(rays, 1.0, True)
    # the dicts carry exactly the reference's keys for these flags (models.py:735-748, 280-284)
    if "history_keys" in g.files:
        assert sorted(history[-1].keys()) == [str(k) for k in g["history_keys"]]
        assert sorted(renderings[-1].keys()) == [str(k) for k in g["rendering_keys"]]
    for L in range(2):
        for k in hist_keys:
            a = g[f"L{L}_h_{k}"]
            tol = 2e-4 if k == "normals_pred" else (1e-4 if k == "density" else 2e-6)
            if chains == "f16x2" and k in ("rgb", "specular"):
                tol = 5e-6                                 # the f16x2 training forward's directional trunk takes x as ONE half (round 5): 2.1e-6
            if L > 0 and k not in ("sdist", "weights"):
                tol = max(tol, 5e-5)                   # level-1 sample positions differ by an ulp
            np.testing.assert_allclose(history[L][k].detach().cpu().numpy().reshape(a.shape), a, rtol=0, atol=tol, err_msg=f"L{L} {k}")
        for k in rend_keys:
            a = g[f"L{L}_r_{k}"]
            x = renderings[L][k].detach().cpu().numpy().reshape(a.shape)
            tol = 5e-6 + (1e-6 / np.maximum(g[f"L{L}_r_acc"], 1e-6) if k == "distance_mean" else 0.0)
            assert np.all(np.abs(x - a) <= tol), (L, k, np.abs(x - a).max())
        assert np.abs(renderings[L]["rgb"].detach().cpu().numpy() - g[f"L{L}_r_rgb"]).max() <= 1e-4
    if not train:
        return
    batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
    total, terms, _ = train_utils.compute_losses(model, batch, rays, renderings, history, cfg)
    assert float(terms["data"].detach()) == pytest.approx(float(g["loss_data"]), rel=1e-5)
    assert float(terms["orientation"].detach()) == pytest.approx(float(g["loss_orientation"]), rel=2e-4)
    if "loss_normal" in g.files:
        assert float(terms["predicted_normals"].detach()) == pytest.approx(float(g["loss_normal"]), rel=6e-4 if posenc else 2e-4)
    else:
        assert "predicted_normals" not in terms and "normals" not in history[0]
    assert float(total.detach()) == pytest.approx(float(g["loss_total"]), rel=1e-5)
    total.backward()
    if flat:
        grads = mlp.flat_parameter().grad.cpu().numpy()
    else:
        grads = np.zeros(mlp.num_params, np.float32)
        for spec, lin in mlp._named_linears():
            assert lin.weight.grad is not None and tuple(lin.weight.grad.shape) == (spec.out_dim, spec.in_dim), spec.name
            grads[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = lin.weight.grad.reshape(-1).cpu().numpy()
            grads[spec.b_off:spec.b_off + spec.out_dim] = lin.bias.grad.cpu().numpy()
    assert grads.shape == (len(idx),)
    ref = g["grads_sub"]
    assert np.linalg.norm(grads[::61] - ref) / np.linalg.norm(ref) < 2e-4
    norms = g["grads_tensor_l2"]
    for i, sp in enumerate(mlp.specs):
        nw = sp.out_dim * sp.in_dim
        assert np.linalg.norm(grads[sp.w_off:sp.w_off + nw]) == pytest.approx(norms[i, 0], rel=2e-3, abs=1e-12), sp.name
    configs.clear_config()


@pytest.mark.parametrize("name", ["model_raydist_reciprocal_eval", "model_raydist_log_eval", "model_raydist_piecewise_eval",
                                  "model_nointegration_eval", "model_raydist_nointegration_train"])
@pytest.mark.parametrize("chains", ["f32", "f16x2"])
def test_raydist_and_disable_integration_vs_reference(hip, name, chains):
    """Model.raydist_fn (coord.construct_ray_warps: reciprocal / log / 'piecewise') and Model.disable_integration through
    cfg.raydist / cfg.disable_integration: Model.__call__ against the reference's outputs, and for the training fixture its
    losses and autograd gradients.  Tolerances of the un-integrated encoding as in the oracle test (level-1 positions differ by
    an ulp, sin(2^15 x) is not attenuated)."""
    import os
    from refnerf_pl_amd import configs, layout, models, train_utils, utils
    g = load_golden(name)
    train = name.endswith("train")
    if chains != "f32" and not train:
        pytest.skip("chain modes: training only")
    noint = bool(int(g["disable_integration"]))
    fn = {"": None, "piecewise": "piecewise", "reciprocal": torch.reciprocal, "log": "@torch.log"}[str(g["raydist_fn"])]
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                            [str(b) for b in g["bindings"] if str(b)]
                                            + [f"Config.hip_train_precision = '{chains}'", f"Config.hip_bwd_precision = '{chains}'"])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV)
    model.raydist_fn, model.disable_integration = fn, noint          # read at call time, as in the reference
    model.nerf_mlp.load_flat_params(params_from_golden(g))
    rays = utils.rays_from_dict(rays_from_golden(g), DEV)
    model.train(train)
    with torch.set_grad_enabled(train):
        renderings, history = model(rays, 1.0, True)
    for L in range(2):
        for k in HIST_KEYS:
            a = g[f"L{L}_h_{k}"]
            tol = 2e-4 if k == "normals_pred" else (1e-4 if k == "density" else 2e-6)
            if L > 0 and k not in ("sdist", "weights"):
                tol = max(tol, 5e-5)
            if L > 0 and noint:
                tol = 5e-5 if k == "sdist" else (2e-4 if k == "weights" else (3e-2 if k in ("density", "normals_pred") else 1e-3))
            np.testing.assert_allclose(history[L][k].detach().cpu().numpy().reshape(a.shape), a, rtol=0, atol=tol, err_msg=f"L{L} {k}")
        for k in REND_KEYS:
            a = g[f"L{L}_r_{k}"]
            x = renderings[L][k].detach().cpu().numpy().reshape(a.shape)
            base = 2e-4 if (noint and L > 0) else 5e-6
            tol = (4 * base + 1e-6 / np.maximum(g[f"L{L}_r_acc"], 1e-6)) if k in ("distance", "distance_mean") else base
            assert np.all(np.abs(x - a) <= tol), (L, k, np.abs(x - a).max())
        assert np.abs(renderings[L]["rgb"].detach().cpu().numpy() - g[f"L{L}_r_rgb"]).max() <= 1e-4
    if train:
        batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
        total, terms, _ = train_utils.compute_losses(model, batch, rays, renderings, history, cfg)
        assert float(terms["data"].detach()) == pytest.approx(float(g["loss_data"]), rel=2e-4)
        assert float(terms["orientation"].detach()) == pytest.approx(float(g["loss_orientation"]), rel=5e-3)
        total.backward()
        grads = torch.cat([p.grad.flatten() for p in model.nerf_mlp.ordered_parameters()]).cpu().numpy()
        ref = g["grads_sub"]
        rel = float(np.linalg.norm(grads[::97] - ref) / np.linalg.norm(ref))
        print(name, "gradient rel-L2 vs reference", rel)
        assert rel < 2e-2
    configs.clear_config()


@pytest.mark.parametrize("n_rays,n_samples", [(2051, 192), (4099, 96), (2050, 160)])
def test_record_ring_kernel(hip, n_rays, n_samples):
    """The 16-bit inference kernel's ring variant (per-sample records in a 512-row ring, rays composited behind the pass
    that completes them: taken when rays_per_wg * N must exceed 640 records to fill whole 256-sample passes and the grid
    has >= 512 workgroups -- 4 x 192, 8 x 96, 8 x 160).  Same rays through both variants: a 600-ray prefix alone takes the
    plain kernel (too few workgroups for the ring), the full ragged batch the ring -- same arithmetic up to the order of the
    compositing sums; the tail rays (partly filled last workgroup) against the f32 mode."""
    import os
    from refnerf_pl_amd import configs, models, synthetic, utils
    rd = synthetic.blender_rays(n_rays, seed=21, center_frac=0.5)
    sub = {k: v[:600] for k, v in rd.items()}
    out = {}
    for prec in ("f32", "bf16", "f16", "f16x2"):
        configs.clear_config()
        configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                                [f"Model.num_prop_samples = {n_samples}", f"Model.num_nerf_samples = {n_samples}",
                                                 f"Config.hip_precision = '{prec}'"])
        model = models.construct_model(utils.dummy_rays(), configs.Config()).to(DEV).eval()
        model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
        with torch.no_grad():
            out[prec] = model(utils.rays_from_dict(rd, DEV), 1.0, True)
            if prec != "f32":
                out[prec + "_sub"] = model(utils.rays_from_dict(sub, DEV), 1.0, True)
    ref_r, _ = out["f32"]
    for prec in ("bf16", "f16", "f16x2"):
        (rend, hist), (rend_s, hist_s) = out[prec], out[prec + "_sub"]
        for L in range(2):
            for k in ("rgb", "diffuse", "specular", "acc", "distance", "normals_pred", "tint", "roughness", "distance_mean"):
                err = float((rend[L][k][:600] - rend_s[L][k]).abs().max())
                assert err <= 2e-6 * (4.0 if "distance" in k else 1.0), (prec, L, k, err)
            assert bool((hist[L]["sdist"][:600] == hist_s[L]["sdist"]).all())          # same resampling, bit for bit
            assert float((hist[L]["weights"][:600] - hist_s[L]["weights"]).abs().max()) <= 1e-6
            for k in ("distance_percentile_5", "distance_median", "distance_percentile_95"):
                assert rend[L][k].dtype == torch.float64
                assert float((rend[L][k][:600] - rend_s[L][k]).abs().max()) <= 1e-5, (prec, L, k)
            # the ragged tail: finite, and the 16-bit distance from the f32 mode that these weights always show
            assert bool(torch.isfinite(rend[L]["rgb"]).all()) and bool(torch.isfinite(rend[L]["distance_median"]).all())
            assert float((rend[L]["rgb"][-16:] - ref_r[L]["rgb"][-16:]).abs().max()) <= 1e-4
    configs.clear_config()


@pytest.mark.parametrize("name", ["model_variant_eval", "model_posenc_eval"])
@pytest.mark.parametrize("prec", ["bf16", "f16", "f16x2"])
def test_nerfmlp_variants_16bit_modes(hip, name, prec):
    """The 16-bit inference kernels take the same embedded image and the same cfg.dir_enc: rendered RGB against the
    reference's vectors (random-init-like weights: the 1e-4 bar holds)."""
    import os
    from helpers import posenc_params, variant_params
    from refnerf_pl_amd import configs, models, utils
    g = load_golden(name)
    _, true_blob, _ = posenc_params(g) if "posenc" in name else variant_params(g)
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                            [str(b) for b in g["bindings"]] + [f"Config.hip_precision = '{prec}'"])
    model = models.construct_model(utils.dummy_rays(), configs.Config()).to(DEV).eval()
    model.nerf_mlp.load_flat_params(true_blob)
    with torch.no_grad():
        renderings, _ = model(utils.rays_from_dict(rays_from_golden(g), DEV), 1.0, False)
    for L in range(2):
        err = np.abs(renderings[L]["rgb"].cpu().numpy() - g[f"L{L}_r_rgb"]).max()
        assert err <= (1e-5 if prec == "f16x2" else 1e-4), (L, err)
    configs.clear_config()


@pytest.mark.parametrize("name", ["model_raydist_reciprocal_eval", "model_raydist_log_eval", "model_raydist_piecewise_eval",
                                  "model_nointegration_eval"])
def test_raydist_and_disable_integration_f16x2(hip, name):
    """the same reference fixtures in the parity-grade 16-bit mode (Config.hip_precision = 'f16x2'): rendered RGB within
    north_star's 1e-4 of the reference (measured ~1e-6; un-integrated encoding: sin(2^15 x) unattenuated)"""
    import os
    from refnerf_pl_amd import configs, models, utils
    g = load_golden(name)
    fn = {"": None, "piecewise": "piecewise", "reciprocal": torch.reciprocal, "log": "@torch.log"}[str(g["raydist_fn"])]
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                            [str(b) for b in g["bindings"] if str(b)] + ["Config.hip_precision = 'f16x2'"])
    model = models.construct_model(utils.dummy_rays(), configs.Config()).to(DEV).eval()
    model.raydist_fn, model.disable_integration = fn, bool(int(g["disable_integration"]))
    model.nerf_mlp.load_flat_params(params_from_golden(g))
    with torch.no_grad():
        renderings, history = model(utils.rays_from_dict(rays_from_golden(g), DEV), 1.0, True)
    for L in range(2):
        err = np.abs(renderings[L]["rgb"].cpu().numpy() - g[f"L{L}_r_rgb"]).max()
        print(name, "f16x2 L%d RGB L-inf vs reference %.2e" % (L, err))
        assert err <= 1e-4, (L, err)
    assert np.array_equal(history[0]["sdist"].cpu().numpy(), g["L0_h_sdist"].reshape(history[0]["sdist"].shape))
    configs.clear_config()


@pytest.mark.parametrize("name,extra", [("model_blender_sharp_train", []),
                                        ("model_llff_linear_train", ["Config.orientation_loss_target = 'normals'"]),
                                        ("model_trained_train", ["Config.predicted_normal_loss_mult = 0.", "Config.predicted_normal_coarse_loss_mult = 0."])])
def test_fused_refnerf_losses_match_train_utils(hip, name, extra):
    """Config.hip_fused_losses: data + orientation + predicted-normal losses of a level through refnerf_losses_forward /
    _backward (one pass each way) against the train_utils path (ATen ops + autograd) on the same forward: loss terms to
    1e-6, parameter gradients to fp32 round-off, and the reference's golden loss values.  Also with the orientation
    target on the (detached) density normals and with a term switched off."""
    import os
    from refnerf_pl_amd import configs, layout, models, train_utils, utils
    g = load_golden(name)
    bindings = [str(b) for b in g["bindings"] if str(b)] + list(extra)
    res = {}
    for fused in (False, True):
        configs.clear_config()
        configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                                bindings + [f"Config.hip_fused_losses = {fused}"])
        cfg = configs.Config()
        model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).train()
        model.nerf_mlp.load_flat_params(params_from_golden(g))
        rays = utils.rays_from_dict(rays_from_golden(g), DEV)
        batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
        renderings, history = model(rays, 1.0, False)
        total, terms, stats = train_utils.compute_losses(model, batch, rays, renderings, history, cfg)
        total.backward()
        flat = torch.zeros(layout.NUM_PARAMS)
        for spec, lin in model.nerf_mlp._named_linears():
            flat[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = lin.weight.grad.reshape(-1).cpu()
            flat[spec.b_off:spec.b_off + spec.out_dim] = lin.bias.grad.cpu()
        res[fused] = (float(total.detach()), {k: float(v.detach()) for k, v in terms.items()}, stats["mses"].cpu().numpy(), flat.numpy())
    (t0, terms0, mses0, g0), (t1, terms1, mses1, g1) = res[False], res[True]
    assert set(terms0) == set(terms1)
    assert t1 == pytest.approx(t0, rel=2e-6)
    for k in terms0:
        assert terms1[k] == pytest.approx(terms0[k], rel=2e-6, abs=1e-12), k
    np.testing.assert_allclose(mses1, mses0, rtol=2e-6)
    rel = np.linalg.norm(g1 - g0) / np.linalg.norm(g0)
    print(f"{name}: fused vs unfused losses: total {t1:.8f} / {t0:.8f}, gradient rel-L2 {rel:.2e}")
    assert rel < 2e-6, rel
    if not extra:
        assert terms1["data"] == pytest.approx(float(g["loss_data"]), rel=1e-5)
        assert terms1["orientation"] == pytest.approx(float(g["loss_orientation"]), rel=2e-4)
        assert terms1["predicted_normals"] == pytest.approx(float(g["loss_normal"]), rel=2e-4)


def test_flat_gradient_mode(hip):
    """Config.hip_flat_grads: the level backward's gradient goes to MLP.flat_parameter().grad (ONE tensor) instead of the
    46 nn.Parameters: bit-identical values, and an Adam step on the flat parameter moves the 46 parameters (views of
    the blob) exactly as an Adam step on the parameters themselves does."""
    import os
    from refnerf_pl_amd import configs, layout, models, train_utils, utils
    g = load_golden("model_blender_sharp_train")
    out = {}
    for flat_mode in (False, True):
        configs.clear_config()
        configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                                [f"Config.hip_flat_grads = {flat_mode}"])
        cfg = configs.Config()
        model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).train()
        model.nerf_mlp.load_flat_params(params_from_golden(g))
        rays = utils.rays_from_dict(rays_from_golden(g), DEV)
        batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
        params = [model.nerf_mlp.flat_parameter()] if flat_mode else list(model.parameters())
        opt = torch.optim.Adam(params, lr=1e-3)
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            renderings, history = model(rays, 1.0, False)
            total, _, _ = train_utils.compute_losses(model, batch, rays, renderings, history, cfg)
            total.backward()
            if flat_mode:
                grad = model.nerf_mlp.flat_parameter().grad.detach().cpu().numpy().copy()
                assert all(p.grad is None for p in model.parameters())
            else:
                grad = np.zeros(layout.NUM_PARAMS, np.float32)
                for spec, lin in model.nerf_mlp._named_linears():
                    grad[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = lin.weight.grad.reshape(-1).cpu().numpy()
                    grad[spec.b_off:spec.b_off + spec.out_dim] = lin.bias.grad.cpu().numpy()
            opt.step()
        # after two steps: blob, state_dict view of it, last gradient, last loss
        sd = model.state_dict()
        out[flat_mode] = (model.nerf_mlp.flat_params().detach().cpu().numpy().copy(), sd["nerf_mlp.rgb.weight"].cpu().numpy().copy(),
                          grad, float(total.detach()))
    assert np.array_equal(out[True][2], out[False][2]), "flat-mode gradient must be bit-identical"
    assert out[True][3] == out[False][3]
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=0, atol=1e-7)      # Adam: foreach over 92 tensors vs one tensor
    np.testing.assert_allclose(out[True][1], out[False][1], rtol=0, atol=1e-7)
    assert np.abs(out[True][0] - params_from_golden(g)).max() > 1e-4                 # the steps did move the weights
    # regression: a FUSED optimiser does not bump the tensors' version counters -- the kernels must still see its update
    # (training forwards re-pack unconditionally; so does the first inference call after them)
    opt = torch.optim.Adam([model.nerf_mlp.flat_parameter()], lr=1e-3, fused=True)
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        renderings, history = model(rays, 1.0, False)
        total, _, _ = train_utils.compute_losses(model, batch, rays, renderings, history, cfg)
        total.backward()
        opt.step()
        losses.append(float(total.detach()))
    assert len(set(losses)) == 3, losses
    model.eval()
    with torch.no_grad():
        e1 = model(rays, 1.0, False)[0][-1]["rgb"].clone()
    model.train()
    opt.zero_grad(set_to_none=True)
    renderings, history = model(rays, 1.0, False)
    train_utils.compute_losses(model, batch, rays, renderings, history, cfg)[0].backward()
    opt.step()
    model.eval()
    with torch.no_grad():
        e2 = model(rays, 1.0, False)[0][-1]["rgb"]
    assert (e1 - e2).abs().max() > 0


def test_train_ray_batcher_on_device(hip):
    """datasets.TrainRayBatcher.next(): the rays of a random batch are what camera_utils.cast_ray_batch gives for the
    drawn pixels (itself pinned by tests/golden/camera.npz), colours gathered on the device, and the batch feeds
    Model.__call__ directly."""
    from refnerf_pl_amd import camera_utils, datasets, synthetic, utils
    rng = np.random.default_rng(0)
    n, H, W = 3, 40, 60
    imgs = rng.random((n, H, W, 3)).astype(np.float32)
    p2c = np.tile(camera_utils.get_pixtocam(80.0, W, H)[None], (n, 1, 1))
    c2w = np.stack([np.concatenate([synthetic._rot(i), (synthetic._rot(i) @ np.array([0., 0., 4.]))[:, None]], 1) for i in range(n)]).astype(np.float32)
    b = datasets.TrainRayBatcher(imgs, (p2c, c2w, None, None), 2., 6., 256, device=DEV, seed=5)
    batch = b.next()
    r = batch.rays
    assert r.origins.shape == (256, 1, 1, 3) and r.origins.is_cuda and batch.rgb.is_cuda
    cam = r.cam_idx[..., 0].long().cpu().numpy()
    # recover the pixels from the image plane and compare with a host-side cast of the same pixels
    pix = utils.Pixels(pix_x_int=None, pix_y_int=None, lossmult=r.lossmult, near=r.near, far=r.far, cam_idx=r.cam_idx)
    ip = r.imageplane.cpu().numpy()
    px = np.rint(ip[..., 0] * 80.0 + W / 2 - 0.5).astype(np.int32)
    py = np.rint(-ip[..., 1] * 80.0 + H / 2 - 0.5).astype(np.int32)
    np.testing.assert_allclose(batch.rgb.cpu().numpy(), imgs[cam, py, px])
    for i in range(0, 256, 37):
        c = int(cam[i, 0, 0])
        o, d, v, rad, _ = synthetic._pixels_to_rays(px[i].reshape(-1), py[i].reshape(-1), 80.0, W, H, c2w[c].astype(np.float64))
        np.testing.assert_allclose(r.origins[i].reshape(-1, 3).cpu().numpy(), o, atol=1e-6)
        np.testing.assert_allclose(r.directions[i].reshape(-1, 3).cpu().numpy(), d, atol=2e-6)
        np.testing.assert_allclose(r.radii[i].reshape(-1).cpu().numpy(), rad.reshape(-1), rtol=1e-4)


def test_edge_shapes_and_errors(hip, O):
    """Ragged / edge sizes: R not a multiple of the workgroup tile, N in {2, 33,
    64, 192, 256}, a single ray; and the reference's error cases."""
    from refnerf_pl_amd import synthetic
    P = synthetic.make_params(1, 0.05, 10.0)
    packed = hip.pack_weights(torch.tensor(P, device=DEV))
    for R, n0, n1 in ((1, 64, 64), (7, 33, 2), (5, 192, 256), (3, 256, 192)):
        rays = synthetic.blender_rays(R, seed=R, center_frac=0.3)
        ref = O.model_forward(P, rays, num_prop_samples=n0, num_nerf_samples=n1)
        outs = run_hip_model(hip, P, rays, {}, dict(num_prop_samples=n0, num_nerf_samples=n1))
        for L in range(2):
            np.testing.assert_allclose(outs[L]["r_rgb"], ref[L]["r_rgb"], rtol=0, atol=1e-4, err_msg=f"R={R} L={L}")
            np.testing.assert_allclose(outs[L]["weights"], ref[L]["weights"], rtol=0, atol=2e-5)
            assert np.mean(outs[L]["bin_idx"] == ref[L]["bin_idx"]) > 0.999
    r = dev_rays(synthetic.blender_rays(2, seed=1))
    sd = torch.tensor([[0.0, 1.0]], device=DEV).repeat(2, 1)
    w = torch.ones((2, 1), device=DEV)
    with pytest.raises(ValueError, match="num_samples must be > 1"):
        hip.level_forward(packed, hip.default_cfg(n_samples=1), r, sd, w)
    with pytest.raises(ValueError, match="ray_shape"):
        hip.level_forward(packed, hip.default_cfg(ray_shape=3), r, sd, w)


@pytest.mark.parametrize("precision", [0, 1])
def test_opaque_background(hip, O, precision):
    """Model.opaque_background (render.py:139-143): last interval alpha = 1, acc = 1, no NaN."""
    from refnerf_pl_amd import synthetic
    P = synthetic.make_params(seed=3, bias_scale=0.05, sharpen=8.0)
    rays = synthetic.blender_rays(7, seed=11, center_frac=0.5)
    kw, lv = dict(opaque_background=1), dict(num_levels=2, num_prop_samples=33, num_nerf_samples=64)
    ref = O.model_forward(P, rays, **lv, **kw)
    outs = run_hip_model(hip, P, rays, kw, lv, precision=precision)
    tol = 1e-5 if precision == 0 else 1e-3
    for res, orc in zip(outs, ref):
        for k in ("weights", "r_rgb", "r_acc", "r_distance"):
            assert np.isfinite(res[k]).all(), k
            if precision == 0 or k != "weights":
                np.testing.assert_allclose(res[k], orc[k].reshape(res[k].shape), rtol=0, atol=tol, err_msg=k)
        np.testing.assert_allclose(res["r_acc"], 1.0, rtol=0, atol=1e-5)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_shapes_vs_oracle(hip, O, seed):
    """Seeded sweep over ragged (R, N_prop, N_fine, levels, ray shape, colour options): both arithmetic
    modes against the oracle (fp32: indices bit-exact, RGB 1e-5; bf16: RGB 2e-4)."""
    from refnerf_pl_amd import synthetic
    rng = np.random.default_rng(100 + seed)
    R = int(rng.integers(1, 41))
    nprop, nfine = int(rng.integers(2, 200)), int(rng.integers(2, 260))
    levels = int(rng.integers(1, 4))
    kw = dict(ray_shape=int(rng.integers(0, 2)), opaque_background=int(rng.integers(0, 2)),
              srgb_mapping=int(rng.integers(0, 2)), render_srgb_mode=int(rng.integers(0, 5)))
    lv = dict(num_levels=levels, num_prop_samples=nprop, num_nerf_samples=nfine)
    P = synthetic.make_params(seed=seed, bias_scale=0.05, sharpen=float(rng.choice([1.0, 8.0, 20.0])))
    rays = (synthetic.blender_rays if seed % 2 == 0 else synthetic.llff_rays)(R, seed=seed + 3)
    ref = O.model_forward(P, rays, **lv, **kw)
    f32 = run_hip_model(hip, P, rays, kw, lv, precision=0)
    bf = run_hip_model(hip, P, rays, kw, lv, precision=1)
    ok = np.ones(R, bool)                 # rays whose sample indices agree with the oracle at every level so far
    for L in range(levels):
        eq = f32[L]["bin_idx"] == ref[L]["bin_idx"]
        if L == 0:
            assert eq.all()               # identical inputs: the sampler itself is bit-exact
        else:
            # deeper levels resample from the previous level's weights, which differ from the oracle's by
            # MLP summation order (1e-7): a quantile within an ulp of a CDF knot may take the neighbouring bin
            assert eq.mean() >= 0.999, (L, R, nprop, nfine, kw, eq.mean())
        ok &= eq.all(-1)
        for k in ("r_rgb", "r_acc", "r_distance", "weights"):
            a, b = f32[L][k], ref[L][k].reshape(f32[L][k].shape)
            np.testing.assert_allclose(a[ok], b[ok], rtol=0, atol=1e-5, err_msg=f"f32 L{L} {k}")
        assert np.isfinite(f32[L]["r_rgb"]).all() and np.isfinite(bf[L]["r_rgb"]).all()
    # bf16: compare the final level only where its CDF bins agree with the fp32 mode (a flipped bin moves a sample)
    same = ok.copy()
    for L in range(1, levels):
        same &= (bf[L]["bin_idx"] == f32[L]["bin_idx"]).all(-1)
    if same.any():
        np.testing.assert_allclose(bf[-1]["r_rgb"][same], ref[-1]["r_rgb"][same], rtol=0, atol=2e-4)


def test_full_size_properties(hip):
    """BASELINE configs[1] size (4096 x 128 x 2): size-independent properties --
    weights >= 0, acc = sum(w) <= 1, sdist monotone in [0,1], level-0 sampling
    exactly uniform, render = sum(w*c)+bg, determinism (two runs bit-equal)."""
    from refnerf_pl_amd import synthetic
    P = synthetic.make_params(0, 0.05, 20.0)
    rays = synthetic.blender_rays(4096, seed=1, center_frac=0.5)
    a = run_hip_model(hip, P, rays, {}, {})
    b = run_hip_model(hip, P, rays, {}, {})
    for L in range(2):
        for k in a[L]:
            assert np.array_equal(a[L][k], b[L][k]), f"non-deterministic {k}"
        w, sd = a[L]["weights"], a[L]["sdist"]
        assert np.all(np.isfinite(a[L]["r_rgb"]))
        assert w.min() >= 0 and np.all(w.sum(-1) <= 1 + 1e-5)
        assert np.all(np.diff(sd, axis=-1) >= 0) and sd.min() >= 0 and sd.max() <= 1
        np.testing.assert_allclose(a[L]["r_acc"], w.sum(-1), rtol=0, atol=1e-5)
        bg = np.maximum(0, 1 - a[L]["r_acc"])[:, None]
        np.testing.assert_allclose(a[L]["r_rgb"], (w[..., None] * a[L]["rgb"]).sum(1) + bg, rtol=0, atol=2e-5)
    from oracle import oracle as Orc
    u = Orc.linspace_u(128)
    mid = (u[1:] + u[:-1]) / 2
    assert np.array_equal(a[0]["sdist"][0, 1:-1], mid.astype(np.float32))
    assert np.all(a[0]["sdist"] == a[0]["sdist"][0:1])


# ---------------------------------------------------------------- bf16 mode
@pytest.mark.parametrize("precision", [1, 2], ids=["bf16", "f16"])
@pytest.mark.parametrize("name", [n for n in EVAL_CASES if n in STRICT_INDEX_CASES])
def test_bf16_mode_tolerance(hip, name, precision):
    """The 16-bit MFMA modes (bf16 / f16 weights and activations, fp32 accumulate, hardware
    transcendentals) against the reference golden vectors on the random-init fixtures: rendered RGB L-inf
    <= 1e-4 (the north-star bar; measured 1e-5 .. 9e-5 in bf16 and <= 1e-5 in f16, deterministic),
    per-sample colour 2e-3, >= 99% identical CDF bin indices."""
    g = load_golden(name)
    P = params_from_golden(g)
    rays = rays_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    outs = run_hip_model(hip, P, rays, kw, lv, precision=precision)
    f32 = run_hip_model(hip, P, rays, kw, lv, precision=0)
    for L, res in enumerate(outs):
        assert np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max() <= (1e-4 if precision == 1 else 2e-5)
        assert np.abs(res["rgb"] - g[f"L{L}_h_rgb"].reshape(res["rgb"].shape)).max() <= 2e-3
        np.testing.assert_allclose(res["weights"], g[f"L{L}_h_weights"], rtol=0, atol=1e-3)
        np.testing.assert_allclose(res["r_acc"], g[f"L{L}_r_acc"], rtol=0, atol=2e-3)
        np.testing.assert_allclose(res["r_distance_mean"], g[f"L{L}_r_distance_mean"], rtol=0, atol=5e-3)
        assert np.mean(res["bin_idx"] == f32[L]["bin_idx"]) >= 0.99
        assert np.all(np.isfinite(res["r_percentiles"]))


def test_bf16_full_size_properties(hip):
    from refnerf_pl_amd import synthetic
    P = synthetic.make_params(0, 0.05, 20.0)
    rays = synthetic.blender_rays(4096, seed=1, center_frac=0.5)
    a = run_hip_model(hip, P, rays, {}, {}, precision=1)
    b = run_hip_model(hip, P, rays, {}, {}, precision=1)
    ref = run_hip_model(hip, P, rays, {}, {}, precision=0)
    for L in range(2):
        for k in a[L]:
            assert np.array_equal(a[L][k], b[L][k]), f"non-deterministic {k}"
        w, sd = a[L]["weights"], a[L]["sdist"]
        assert w.min() >= 0 and np.all(w.sum(-1) <= 1 + 1e-5)
        assert np.all(np.diff(sd, axis=-1) >= 0) and sd.min() >= 0 and sd.max() <= 1
        np.testing.assert_allclose(a[L]["r_acc"], w.sum(-1), rtol=0, atol=1e-5)
        bg = np.maximum(0, 1 - a[L]["r_acc"])[:, None]
        np.testing.assert_allclose(a[L]["r_rgb"], (w[..., None] * a[L]["rgb"]).sum(1) + bg, rtol=0, atol=2e-5)
        # against the fp32 parity mode at the full BASELINE size
        assert np.abs(a[L]["r_rgb"] - ref[L]["r_rgb"]).max() <= 1e-4
    for R, n0, n1 in ((3, 64, 64), (5, 192, 256), (2, 33, 2)):
        rr = synthetic.blender_rays(R, seed=R, center_frac=0.3)
        x = run_hip_model(hip, P, rr, {}, dict(num_prop_samples=n0, num_nerf_samples=n1), precision=1)
        y = run_hip_model(hip, P, rr, {}, dict(num_prop_samples=n0, num_nerf_samples=n1), precision=0)
        for L in range(2):
            assert np.abs(x[L]["r_rgb"] - y[L]["r_rgb"]).max() <= 2e-3, (R, n0, n1, L)


# ---------------------------------------------------------------- full-size batches against the oracle
def _psnr(a, b):
    return float(-10.0 * np.log10(max(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2), 1e-20)))


def _record(name, payload):
    """Measured parity numbers of this run -> gpurun_out/parity_full_size.json (copied into profiles/ by the builder)."""
    import json
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "parity_full_size.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = payload
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    print(name, payload)


FULL_SIZE = {
    # name: (R, N, param recipe, ray recipe, oracle rays, f32 bound, bf16 bound) -- bounds on rendered RGB L-inf
    "C2_bench_batch": (4096, 128, dict(seed=0, bias_scale=0.05, sharpen=20.0), dict(seed=1, center_frac=0.5), 512, 1e-5, 1e-4),
    "C3_shiny": (8192, 192, dict(seed=0, bias_scale=0.05, sharpen=20.0, roughness_bias=-6.0), dict(seed=1, center_frac=0.5), 512, 1e-5, 1e-4),
    "C2_trained_like": (4096, 128, "trained", dict(seed=3, center_frac=0.8), 512, 3e-5, None),
}


@pytest.mark.parametrize("case", list(FULL_SIZE))
def test_full_size_batches_vs_oracle(hip, O, case):
    """The batches bench.py times (C2, C3) and a C2-sized batch on the trained-like weights, at FULL size on the HIP
    path in both arithmetic modes; the first 512 rays of each also through the CPU oracle (rays are independent, so
    a prefix of the batch is the same computation).  f32 mode: rendered RGB L-inf <= 1e-5 (3e-5 on the trained-like
    network, whose level-1 outputs amplify the ulp-level differences of the level-1 sample positions) and >= 99.9 %
    identical CDF bin indices; bf16 / f16 modes: the north-star bar 1e-4 on the random-init networks, and whatever
    they measure on the trained-like one (recorded; the f32 mode is the parity mode of record there)."""
    from helpers import trained_blob
    from refnerf_pl_amd import synthetic
    R, N, pk, rk, n_or, tol32, tol16 = FULL_SIZE[case]
    P = trained_blob() if pk == "trained" else synthetic.make_params(**pk)
    rays = synthetic.blender_rays(R, **rk)
    lv = dict(num_prop_samples=N, num_nerf_samples=N)
    f32 = run_hip_model(hip, P, rays, {}, lv, precision=0)
    bf = run_hip_model(hip, P, rays, {}, lv, precision=1)
    hf = run_hip_model(hip, P, rays, {}, lv, precision=2)
    sub = {k: v[:n_or] for k, v in rays.items()}
    ref = O.model_forward(P, sub, **lv)
    rec = {"rays": R, "samples": N, "oracle_rays": n_or}
    for L in range(2):
        for tag, out in (("f32", f32), ("bf16", bf), ("f16", hf)):
            a = out[L]
            # size-independent properties on the whole batch
            w, sd = a["weights"], a["sdist"]
            assert np.isfinite(a["r_rgb"]).all()
            assert w.min() >= 0 and np.all(w.sum(-1) <= 1 + 1e-5)
            assert np.all(np.diff(sd, axis=-1) >= 0) and sd.min() >= 0 and sd.max() <= 1
            bg = np.maximum(0, 1 - a["r_acc"])[:, None]
            np.testing.assert_allclose(a["r_rgb"], (w[..., None] * a["rgb"]).sum(1) + bg, rtol=0, atol=3e-5)
            # against the oracle on the prefix
            err = float(np.abs(a["r_rgb"][:n_or] - ref[L]["r_rgb"]).max())
            same = float(np.mean(a["bin_idx"][:n_or] == ref[L]["bin_idx"]))
            rec[f"L{L}_{tag}_rgb_linf_vs_oracle"] = err
            rec[f"L{L}_{tag}_bin_idx_agreement"] = same
            rec[f"L{L}_{tag}_psnr_vs_oracle_db"] = _psnr(a["r_rgb"][:n_or], ref[L]["r_rgb"])
        for tag, out in (("bf16", bf), ("f16", hf)):
            rec[f"L{L}_{tag}_rgb_linf_vs_f32_full_batch"] = float(np.abs(out[L]["r_rgb"] - f32[L]["r_rgb"]).max())
            rec[f"L{L}_{tag}_psnr_vs_f32_db"] = _psnr(out[L]["r_rgb"], f32[L]["r_rgb"])
        rec[f"L{L}_density_max"] = float(f32[L]["density"].max())
    _record(case, rec)
    for L in range(2):
        assert rec[f"L{L}_f32_rgb_linf_vs_oracle"] <= tol32, rec
        assert rec[f"L{L}_f32_bin_idx_agreement"] >= 0.9999, rec      # 65536+ indices: measured >= 99.997 % (CDF ties one ulp apart); 0.999 would hide a 250x regression
        if tol16 is not None:
            assert rec[f"L{L}_bf16_rgb_linf_vs_oracle"] <= tol16, rec
            assert rec[f"L{L}_f16_rgb_linf_vs_oracle"] <= tol16 / 5, rec
        else:
            # trained-like weights: the 16-bit modes do NOT meet the 1e-4 bar (measured: bf16 6e-3 / 5e-2 at level 0 / 1,
            # f16 5e-4 / 1.2e-2; PSNR vs the f32 mode 52-66 dB and 68-85 dB) -- f32 is the parity mode of record; the
            # bounds below only catch regressions of these figures
            assert rec[f"L{L}_bf16_rgb_linf_vs_oracle"] <= 0.1 and rec[f"L{L}_bf16_psnr_vs_f32_db"] >= 48.0, rec
            assert rec[f"L{L}_f16_rgb_linf_vs_oracle"] <= 0.03 and rec[f"L{L}_f16_psnr_vs_f32_db"] >= 64.0, rec
        assert rec[f"L{L}_bf16_bin_idx_agreement"] >= 0.98, rec


@pytest.mark.parametrize("name", ["model_shiny_eval", "model_trained_eval"])
def test_bf16_mode_on_shiny_and_trained_fixtures(hip, name):
    """bf16 mode against the REFERENCE's outputs on the two hard fixtures: the shiny network (roughness ~ 1e-3: the
    degree-16 IDE terms reach the bf16 GEMM unattenuated, through hardware sin/cos) at 192 samples, and the
    trained-like weights, in all three arithmetic modes.  Measured values are recorded; on the shiny network the 16-bit
    modes meet the north-star 1e-4, on the trained-like weights only the f32 mode does (bf16 ~3e-3, f16 ~4e-4 on this
    fixture: 16-bit products in pre-activations of magnitude 20-30) -- the bounds there only catch regressions."""
    g = load_golden(name)
    P = params_from_golden(g)
    rays = rays_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    outs = {tag: run_hip_model(hip, P, rays, kw, lv, precision=prec) for tag, prec in (("f32", 0), ("bf16", 1), ("f16", 2))}
    rec = {}
    for L in range(2):
        for tag, out in outs.items():
            res = out[L]
            rec[f"L{L}_{tag}_rgb_linf_vs_reference"] = float(np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max())
            rec[f"L{L}_{tag}_sample_rgb_linf"] = float(np.abs(res["rgb"] - g[f"L{L}_h_rgb"].reshape(res["rgb"].shape)).max())
            rec[f"L{L}_{tag}_bin_idx_vs_f32"] = float(np.mean(res["bin_idx"] == outs["f32"][L]["bin_idx"]))
            assert np.all(np.isfinite(res["r_percentiles"]))
    _record("fixture_" + name, rec)
    for L in range(2):
        assert rec[f"L{L}_f32_rgb_linf_vs_reference"] <= 1e-5, rec
        assert rec[f"L{L}_bf16_rgb_linf_vs_reference"] <= (1e-4 if "shiny" in name else 2e-2), rec
        assert rec[f"L{L}_f16_rgb_linf_vs_reference"] <= (2e-5 if "shiny" in name else 5e-3), rec


def test_render_image_lean_history_is_invisible(hip):
    """render_image drops the chunk's ray_history; inside it Model.__call__ does not materialise it
    (models.lean_ray_history): identical renderings, None history entries, normal behaviour outside."""
    import os
    from refnerf_pl_amd import configs, models, synthetic, utils
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                            ["Config.render_chunk_size = 96"])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).eval()
    model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
    rd = synthetic.blender_rays(16 * 12, seed=3)
    img = utils.rays_from_dict({k: v.reshape(16, 12, -1) for k, v in rd.items()}, DEV)
    seen = []

    def fn(r):
        out = model(r, 1.0, True)
        seen.append(out[1])
        return out
    with torch.no_grad():
        rendering = models.render_image(fn, img, cfg, verbose=False, device=DEV)
        full, hist = model(utils.rays_from_dict(rd, DEV), 1.0, True)
    assert seen[0][-1]["density"] is None and seen[0][-1]["rgb"] is not None and seen[0][-1]["weights"] is not None
    assert hist[-1]["density"] is not None and hist[-1]["tint"] is not None          # outside: the full history
    for k in ("rgb", "diffuse", "specular", "acc", "distance_mean", "normals_pred", "roughness"):
        assert torch.equal(rendering[k].reshape(full[-1][k].shape), full[-1][k]), k
    assert len(rendering["ray_rgbs"]) == 2 and rendering["ray_rgbs"][0].shape[-1] == 3


def test_dilation_and_anneal_model_options(hip):
    """Model(dilation_bias, dilation_multiplier, anneal_slope) at train_frac 0.3 end to end on the HIP path
    against the reference (the dilation runs as host torch ops in front of the level kernel)."""
    import os
    from refnerf_pl_amd import configs, models, utils
    g = load_golden("model_dilation_anneal_eval")
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                            [str(b) for b in g["bindings"]])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).eval()
    assert model.dilation_bias == 0.0025 and model.anneal_slope == 10.
    model.nerf_mlp.load_flat_params(params_from_golden(g))
    with torch.no_grad():
        rend, hist = model(utils.rays_from_dict(rays_from_golden(g), DEV), float(g["train_frac"]), True)
    np.testing.assert_array_equal(hist[0]["sdist"].cpu().numpy(), g["L0_h_sdist"])
    sd1 = hist[1]["sdist"].cpu().numpy()
    ok = np.abs(sd1 - g["L1_h_sdist"]).max(-1) < 2e-6
    assert ok.mean() >= 0.9                                   # a renormalisation sum 1 ulp apart can move a knot
    np.testing.assert_allclose(rend[1]["rgb"].cpu().numpy()[ok], g["L1_r_rgb"][ok], atol=1e-5)
    np.testing.assert_allclose(rend[1]["rgb"].cpu().numpy(), g["L1_r_rgb"], atol=1e-3)
    np.testing.assert_allclose(rend[0]["rgb"].cpu().numpy(), g["L0_r_rgb"], atol=1e-5)
    big = models.Model(config=cfg, num_prop_samples=192, num_nerf_samples=192, dilation_bias=0.0025, num_levels=2,
                       single_mlp=True, resample_padding=0.01, anneal_slope=0.).to(DEV).eval()
    with pytest.raises(ValueError), torch.no_grad():            # 3 * 192 - 2 intervals > 512
        big(utils.rays_from_dict(rays_from_golden(g), DEV), 1.0, False)


def test_wgrad_modes_agree(hip):
    """cfg.wgrad_mode: the split-bf16 weight-gradient GEMM (default) against the fp32-MFMA one on the same
    saved activations / deltas: fp32-level agreement, both bit-reproducible, unknown modes rejected."""
    from refnerf_pl_amd import _hip, synthetic
    R, N = 37, 96
    P = torch.tensor(synthetic.make_params(0, 0.05, 20.0), device=DEV)
    rays = dev_rays(synthetic.blender_rays(R, seed=4, center_frac=0.4))
    packed = _hip.pack_weights(P, precision=0)
    sd = torch.tensor([[0.0, 1.0]], device=DEV).repeat(R, 1)
    w = torch.ones((R, 1), device=DEV)
    g = torch.Generator(device="cpu").manual_seed(0)
    g_rgb = torch.randn((R, 3), generator=g).to(DEV) * 1e-2
    g_w = torch.randn((R, N), generator=g).to(DEV) * 1e-2
    g_np = torch.randn((R, N, 3), generator=g).to(DEV) * 1e-2
    out = {}
    for mode in (_hip.WGRAD_F32, _hip.WGRAD_BF16X3):
        cfg = _hip.default_cfg(n_samples=N, n_in=1, training=1, compute_extras=0, wgrad_mode=mode)
        for rep in range(2):
            res = _hip.level_forward(packed, cfg, rays, sd, w, history=True, save_activations=True)
            grads = torch.zeros(_hip.NUM_PARAMS, device=DEV)
            _hip.level_backward(packed, cfg, rays, res, g_rgb, g_w, g_np, grads)
            out[(mode, rep)] = grads.cpu().double()
        assert torch.equal(out[(mode, 0)], out[(mode, 1)])                 # no atomics: bit-reproducible
    a, b = out[(_hip.WGRAD_F32, 0)], out[(_hip.WGRAD_BF16X3, 0)]
    assert float(a.norm()) > 0
    # 2^-16 per product (the dropped lo*lo term); these random-sign seeds cancel more than a real loss does
    # (2.4e-7 on the C2 training step, scripts/ab_wgrad.py)
    assert float((a - b).norm() / a.norm()) < 1e-5
    assert float((a - b).abs().max()) < 1e-5 * float(a.abs().max())
    bad = _hip.default_cfg(n_samples=N, n_in=1, training=1, compute_extras=0, wgrad_mode=7)
    res = _hip.level_forward(packed, bad, rays, sd, w, history=True, save_activations=True)
    with pytest.raises(ValueError):
        _hip.level_backward(packed, bad, rays, res, g_rgb, g_w, g_np, torch.zeros(_hip.NUM_PARAMS, device=DEV))


@pytest.mark.parametrize("chains", ["f32", "f16x2"])
def test_c_abi_training_without_torch(hip, tmp_path, chains):
    """examples/c_abi_train_demo.cpp: a training step (two refnerf_level_forward_train calls, the data loss by hand,
    two refnerf_level_backward calls into one gradient blob) from a C++ host that links only the C ABI; loss and
    gradient equal the Python host's (Model.__call__ autograd nodes + train_utils.compute_data_loss)."""
    import os
    import re
    import subprocess
    import __graft_entry__ as ge
    from refnerf_pl_amd import camera_utils, configs, layout, models, synthetic, train_utils, utils
    exe = ge.build_c_demo("c_abi_train_demo")
    W, H, focal = 12, 9, 20.0
    blob = synthetic.make_params(seed=2, bias_scale=0.05, sharpen=20.0)
    c2w, _ = synthetic.blender_camera(seed=3)
    gt = synthetic.target_rgb(W * H, seed=5)
    blob.astype(np.float32).tofile(tmp_path / "w.f32")
    c2w.astype(np.float32).tofile(tmp_path / "c2w.f32")
    gt.astype(np.float32).tofile(tmp_path / "gt.f32")
    out = subprocess.run([exe, str(tmp_path / "w.f32"), str(tmp_path / "c2w.f32"), str(W), str(H), str(focal),
                          str(tmp_path / "gt.f32"), str(tmp_path / "g.f32"), {"f32": "0", "f16x2": "3"}[chains]],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    g_c = np.fromfile(tmp_path / "g.f32", np.float32)
    loss_c = float(re.search(r"loss ([0-9.eE+-]+),", out.stdout).group(1))
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                            ["Model.num_prop_samples = 48", "Model.num_nerf_samples = 48",
                                             "Config.orientation_loss_mult = 0.", "Config.orientation_coarse_loss_mult = 0.",
                                             "Config.predicted_normal_loss_mult = 0.", "Config.predicted_normal_coarse_loss_mult = 0.",
                                             f"Config.hip_train_precision = '{chains}'", f"Config.hip_bwd_precision = '{chains}'"])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).train()
    model.nerf_mlp.load_flat_params(blob)
    rays = camera_utils.cast_pinhole_rays(c2w.astype(np.float32), H, W, focal, 2.0, 6.0, device=torch.device(DEV)).reshape(H * W, -1)
    rend, hist = model(rays, 1.0, False)
    total, terms, _ = train_utils.compute_losses(model, utils.Batch(rays=rays, rgb=gt), rays, rend, hist, cfg)
    assert sorted(terms) == ["data"]
    total.backward()
    g_py = np.zeros(layout.NUM_PARAMS, np.float32)
    for spec, lin in model.nerf_mlp._named_linears():
        g_py[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = lin.weight.grad.reshape(-1).cpu().numpy()
        g_py[spec.b_off:spec.b_off + spec.out_dim] = lin.bias.grad.cpu().numpy()
    assert loss_c == pytest.approx(float(total.detach()), rel=1e-5)
    assert np.linalg.norm(g_py) > 0
    assert np.linalg.norm(g_c - g_py) / np.linalg.norm(g_py) < 1e-5


def test_bf16_chain_backward_mode(hip):
    """refnerf_level_backward with cfg.precision = BF16 (Config.hip_bwd_precision = 'bf16'): the transposed GEMM chains
    on bf16 MFMA with deltas rounded to bf16 once per layer; forward, masks, head recompute and the weight-gradient GEMM
    unchanged.  Gradients within bf16 accuracy of the f32 mode, every tensor; bit-reproducible."""
    from refnerf_pl_amd import _hip, layout, synthetic
    R, N = 48, 96
    P = torch.tensor(synthetic.make_params(0, 0.05, 20.0), device=DEV)
    rays = dev_rays(synthetic.blender_rays(R, seed=4, center_frac=0.4))
    packed = _hip.pack_weights(P, precision=0)
    sd = torch.tensor([[0.0, 1.0]], device=DEV).repeat(R, 1)
    w = torch.ones((R, 1), device=DEV)
    g = torch.Generator(device="cpu").manual_seed(0)
    g_rgb = torch.randn((R, 3), generator=g).to(DEV) * 1e-2
    g_w = torch.randn((R, N), generator=g).to(DEV) * 1e-2
    g_np = torch.randn((R, N, 3), generator=g).to(DEV) * 1e-2
    out = {}
    for prec in (_hip.PREC_F32, _hip.PREC_BF16):
        for rep in range(2):
            cfg = _hip.default_cfg(n_samples=N, n_in=1, training=1, compute_extras=0)
            res = _hip.level_forward(packed, cfg, rays, sd, w, history=True, save_activations=True)
            cfg.precision = prec
            grads = torch.zeros(_hip.NUM_PARAMS, device=DEV)
            _hip.level_backward(packed, cfg, rays, res, g_rgb, g_w, g_np, grads)
            out[(prec, rep)] = grads.cpu().double()
        assert torch.equal(out[(prec, 0)], out[(prec, 1)])
    a, b = out[(_hip.PREC_F32, 0)], out[(_hip.PREC_BF16, 0)]
    assert float((a - b).norm() / a.norm()) < 5e-3
    for s in layout.PARAM_SPECS:
        sl = slice(s.w_off, s.w_off + s.out_dim * s.in_dim)
        assert float((a[sl] - b[sl]).norm()) <= 2e-2 * float(a[sl].norm()) + 1e-9, s.name
    # the rgb layer sits above the chains: its gradient only sees the rounding of its 3 delta rows to the bf16 DELTA format
    rgbw = slice(layout.PARAM_SPECS[-1].w_off, layout.PARAM_SPECS[-1].w_off + 3 * 256)
    assert float((a[rgbw] - b[rgbw]).norm()) <= 2e-3 * float(a[rgbw].norm())


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_bf16_chain_training_forward(hip, O, name):
    """Training forward with cfg.precision = BF16 (Config.hip_train_precision = 'bf16'): the fp32-structure kernel with
    its MLP chains on bf16 MFMA.  Level-0 sample indices bit-exact (the resampler does not depend on the MLP), RGB within
    1e-4 of the oracle and of the reference's golden vectors, history within bf16 tolerances, density normals sane."""
    g = load_golden(name)
    P = params_from_golden(g)
    rays = rays_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    ref = O.model_forward(P, rays, training=1, **lv, **kw)
    outs = run_hip_model(hip, P, rays, dict(kw, training=1), lv, precision=1)
    assert np.mean(outs[0]["bin_idx"] == ref[0]["bin_idx"]) == 1.0
    # trained-like weights: bf16 products in pre-activations of magnitude 20-30 -- the bf16 chains measure 4e-3 there
    # (the same as the bf16 eval mode, see test_full_size_batches_vs_oracle); the f32 chains are the parity mode
    rgb_tol, w_tol = (2e-2, 5e-2) if name.startswith("model_trained") else (1e-4, 2e-3)
    for L, (res, orc) in enumerate(zip(outs, ref)):
        err = np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max()
        print(f"{name} bf16 chains L{L}: RGB L-inf vs reference {err:.2e}")
        assert err <= rgb_tol
        # level 1 resamples from level-0 weights that carry bf16 noise: a quantile next to a CDF knot may take the
        # neighbouring bin (the same happens in the bf16 eval mode); per-sample quantities are compared on the other rays
        assert np.mean(res["bin_idx"] == orc["bin_idx"]) > 0.97
        same = (res["bin_idx"] == orc["bin_idx"]).all(-1)
        assert same.any()
        assert np.abs(res["r_rgb"][same] - orc["r_rgb"][same]).max() <= rgb_tol
        assert np.abs(res["weights"][same] - orc["weights"][same]).max() <= w_tol
        nrm = res["normals"][same]
        assert np.abs(np.linalg.norm(nrm, axis=-1) - 1.0).max() < 1e-3
        cos = (nrm * orc["normals"].reshape(res["normals"].shape)[same]).sum(-1)
        # bf16 VJP: tight in the bulk, loose where the density gradient is tiny (ill-conditioned normalisation)
        # (median angular error of the density normals <= 2.5 degrees)
        assert np.median(cos) > 0.999 and np.mean(cos > 0.99) > 0.8 and np.mean(cos > 0.9) > 0.95


@pytest.mark.parametrize("n_rays,n_prop,n_nerf", [(70, 64, 96), (37, 40, 72)])
def test_bf16_chain_training_step(hip, n_rays, n_prop, n_nerf):
    """Whole training step with Config.hip_train_precision = hip_bwd_precision = 'bf16' against the all-f32 step:
    same loss to 1e-4, gradient within bf16 accuracy, rendering within 1e-4.  The second shape is ragged on purpose:
    sample counts that fill neither a 128-sample pass nor the 32-sample chunks of the sample-major block."""
    import os
    from refnerf_pl_amd import configs, models, synthetic, train_utils, utils
    res = {}
    for mode, bwd_mode in (("f32", "f32"), ("bf16", "bf16"), ("bf16", "f32")):
        configs.clear_config()
        configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                                [f"Model.num_prop_samples = {n_prop}", f"Model.num_nerf_samples = {n_nerf}",
                                                 f"Config.hip_train_precision = '{mode}'", f"Config.hip_bwd_precision = '{bwd_mode}'"])
        cfg = configs.Config()
        model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).train()
        model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
        rays = utils.rays_from_dict(synthetic.blender_rays(n_rays, seed=9, center_frac=0.4), DEV)
        batch = utils.Batch(rays=rays, rgb=synthetic.target_rgb(n_rays, seed=3))
        rend, hist = model(rays, 1.0, True)
        total, _, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
        total.backward()
        grad = torch.cat([p.grad.flatten() for p in model.nerf_mlp.ordered_parameters()]).double().cpu()
        res[mode, bwd_mode] = (float(total.detach()), grad, rend[1]["rgb"].detach().cpu())
    (la, ga, ra), (lb, gb, rb) = res["f32", "f32"], res["bf16", "bf16"]
    assert lb == pytest.approx(la, rel=1e-4)
    assert float((ra - rb).abs().max()) <= 1e-4
    assert float((ga - gb).norm() / ga.norm()) < 1e-2
    lc, gc, rc = res["bf16", "f32"]                     # bf16 forward (bf16 ACT rows + sample-major block) into the f32 backward
    assert lc == lb and bool((rc == rb).all())
    assert float((ga - gc).norm() / ga.norm()) < 1e-2
    with pytest.raises(ValueError):
        configs.clear_config()
        configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                                ["Config.hip_bwd_precision = 'fp8'"])
        m = models.construct_model(utils.dummy_rays(), configs.Config()).to(DEV).train()
        m(rays, 1.0, False)

@pytest.mark.parametrize("name", ["model_blender_sharp_train", "model_llff_linear_train", "model_shiny_train", "model_trained_train"])
def test_bf16_chain_training_step_vs_reference(hip, name):
    """The throughput training mode (Config.hip_train_precision = hip_bwd_precision = 'bf16') against the REFERENCE's own
    losses and autograd gradients (the golden training fixtures), not only against this build's f32 mode: what it measures
    is recorded (gpurun_out/parity_full_size.json -> profiles/) and bounded.  On random-init-like networks the gradient is
    within 2e-2 relative L2 (measured 0.5e-2 .. 1.1e-2) and the loss within 2e-4; on the trained-like network the bf16 chains are visibly off (the f32
    chains are the parity mode: test_training_step_gradients)."""
    import os
    from refnerf_pl_amd import configs, layout, models, train_utils, utils
    g = load_golden(name)
    bindings = [str(b) for b in g["bindings"] if str(b)]
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                            bindings + ["Config.hip_train_precision = 'bf16'", "Config.hip_bwd_precision = 'bf16'"])
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).train()
    model.nerf_mlp.load_flat_params(params_from_golden(g))
    rays = utils.rays_from_dict(rays_from_golden(g), DEV)
    batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
    rend, hist = model(rays, 1.0, False)
    total, terms, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
    total.backward()
    grads = torch.cat([p.grad.flatten() for p in model.nerf_mlp.ordered_parameters()]).cpu().numpy()
    ref = g["grads_sub"]
    rel = float(np.linalg.norm(grads[::97] - ref) / np.linalg.norm(ref))
    lrel = abs(float(total.detach()) - float(g["loss_total"])) / abs(float(g["loss_total"]))
    rgb = float(np.abs(rend[1]["rgb"].detach().cpu().numpy() - g["L1_r_rgb"]).max())
    tn = g["grads_tensor_l2"]
    worst = max(abs(np.linalg.norm(grads[s.w_off:s.w_off + s.out_dim * s.in_dim]) / tn[i, 0] - 1.0) for i, s in enumerate(layout.PARAM_SPECS))
    print(f"{name} bf16 chains vs reference: gradient rel-L2 {rel:.2e}, worst tensor-norm error {worst:.2e}, loss rel {lrel:.2e}, RGB L-inf {rgb:.2e}")
    _record("bf16_chain_training_vs_reference/" + name, dict(grad_rel_l2=rel, worst_tensor_norm_err=float(worst), loss_rel=lrel, rgb_linf=rgb))
    trained = name.startswith("model_trained")
    assert rel < (0.25 if trained else 2e-2) and lrel < (5e-2 if trained else 2e-4) and rgb < (5e-2 if trained else 1e-4)
    configs.clear_config()


def test_bf16_chain_training_sample_limit(hip):
    """The bf16-chain training forward keeps a 24 KB weight-stream ring in LDS on top of the level's tiles: it takes
    n_samples <= 294 (f32: 561) and says so instead of running something else."""
    import os
    from refnerf_pl_amd import configs, models, synthetic, utils
    rays = utils.rays_from_dict(synthetic.blender_rays(8, seed=2, center_frac=0.4), DEV)
    for mode, ok in (("f32", True), ("bf16", False)):
        configs.clear_config()
        configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                                ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 320",
                                                 f"Config.hip_train_precision = '{mode}'", f"Config.hip_bwd_precision = '{mode}'"])
        model = models.construct_model(utils.dummy_rays(), configs.Config()).to(DEV).train()
        model.nerf_mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
        if ok:
            rend, _ = model(rays, 1.0, True)
            assert bool(torch.isfinite(rend[1]["rgb"]).all())
        else:
            with pytest.raises(ValueError, match="LDS budget"):
                model(rays, 1.0, True)
    configs.clear_config()


# ---------------------------------------------------------------- enable_pred_specular_density / render_with_specular_density
@pytest.mark.parametrize("name", ["model_specdens_eval", "model_specdens_train"])
@pytest.mark.parametrize("mode", ["f32", "f16x2"])
def test_specular_density_head_vs_reference(hip, name, mode):
    """`NerfMLP.enable_pred_specular_density = True` + `Config.render_with_specular_density = True` (internal/models.py:250-258,
    502-503,583-584,624-625,745-746) against the reference's own outputs: ray_history gains `specular_density` =
    softplus(raw_specular_density(x) + density_bias) between 'specular' and 'roughness', everything else is unchanged (the
    reference computes `specular_weights` and drops them), the state_dict gains the head's two tensors, and in training the head
    receives NO gradient.  Served by a second launch on a weight image whose density head is the extra head
    (MLP.packed_specular_weights)."""
    import os
    from helpers import HIST_KEYS, REND_KEYS
    from refnerf_pl_amd import configs, models, train_utils, utils
    g = load_golden(name)
    train = name.endswith("train")
    bindings = [str(b) for b in g["bindings"]] + [f"Config.hip_precision = '{mode}'", f"Config.hip_train_precision = '{mode}'",
                                                  f"Config.hip_bwd_precision = '{mode}'"]
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")], bindings)
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV)
    mlp = model.nerf_mlp
    assert list(mlp.state_dict().keys()) == [str(k) for k in g["state_dict_keys"]]          # names AND order of the reference's module
    mlp.load_flat_params(params_from_golden(g))
    with torch.no_grad():
        mlp.raw_specular_density.weight.copy_(torch.tensor(g["specdens_w"]))
        mlp.raw_specular_density.bias.copy_(torch.tensor(g["specdens_b"]))
    rays = utils.rays_from_dict(rays_from_golden(g), DEV)
    model.train(train)
    with torch.set_grad_enabled(train):
        renderings, history = model(rays, 1.0, True)
    assert list(history[0].keys()) == [str(k) for k in g["hist_keys"]]
    loose = mode != "f32"
    for L in range(2):
        sd = history[L]["specular_density"].detach().cpu().numpy()
        ref = g[f"L{L}_h_specular_density"]
        assert sd.shape == ref.shape
        # (level 1: sample positions an ulp apart, as for `density`)
        np.testing.assert_allclose(sd, ref, rtol=0, atol=(1e-4 if L else 2e-6) * (3 if loose else 1), err_msg=f"L{L} specular_density")
        np.testing.assert_allclose(history[L]["density"].detach().cpu().numpy(), g[f"L{L}_h_density"], rtol=0, atol=(1e-4 if L else 2e-6) * (3 if loose else 1))
        for k in REND_KEYS:
            if f"L{L}_r_{k}" in g.files:
                a = g[f"L{L}_r_{k}"]
                x = renderings[L][k].detach().cpu().numpy().reshape(a.shape)
                tol = (5e-6 if not loose else 1e-4) + (1e-6 / np.maximum(g[f"L{L}_r_acc"], 1e-6) if k == "distance_mean" else 0.0)
                assert np.all(np.abs(x - a) <= tol), (L, k, float(np.abs(x - a).max()))
    if not train:
        # the stage entry too (MLP.__call__ on caller-supplied Gaussians): same extra key, same values as the level's fixture at
        # level 0 are not available there -- check the key order and that it equals a direct evaluation of the head's formula
        return
    batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
    total, terms, _ = train_utils.compute_losses(model, batch, rays, renderings, history, cfg)
    assert float(total.detach()) == pytest.approx(float(g["loss_total"]), rel=1e-5 if not loose else 2e-5)
    total.backward()
    assert mlp.raw_specular_density.weight.grad is None and mlp.raw_specular_density.bias.grad is None      # as in the reference
    assert bool(g["specdens_grad_is_none"].all())
    from refnerf_pl_amd import layout
    grads = np.zeros(layout.NUM_PARAMS, np.float32)
    for spec, lin in mlp._named_linears():
        grads[spec.w_off:spec.w_off + spec.out_dim * spec.in_dim] = lin.weight.grad.reshape(-1).cpu().numpy()
        grads[spec.b_off:spec.b_off + spec.out_dim] = lin.bias.grad.cpu().numpy()
    rel = float(np.linalg.norm(grads[::97] - g["grads_sub"]) / np.linalg.norm(g["grads_sub"]))
    print(f"{name} [{mode}]: gradient rel-L2 vs reference {rel:.2e}")
    assert rel < 2e-4
    configs.clear_config()


def test_specular_density_stage_entry_and_errors(hip):
    """MLP.__call__ (the stage entry on caller-supplied Gaussians) with the head: the extra key sits where the reference puts it and
    equals `density` of the same module whose density head carries the extra head's parameters; the reference's ValueErrors."""
    import os
    from refnerf_pl_amd import configs, models, synthetic, utils
    gin = os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")
    configs.clear_config()
    configs.parse_config_files_and_bindings([gin], ["Config.render_with_specular_density = True"])
    model = models.construct_model(utils.dummy_rays(), configs.Config()).to(DEV).eval()      # the flag without the head
    with pytest.raises(ValueError, match="Specular density prediction from mlps should be enabled"):   # models.py:250-252
        with torch.no_grad():
            model(utils.rays_from_dict(dict(synthetic.blender_rays(8, seed=1, center_frac=0.4)), DEV), 1.0, False)
    configs.clear_config()
    configs.parse_config_files_and_bindings([gin], [])
    rng = np.random.default_rng(5)
    mlp = models.NerfMLP(enable_pred_specular_density=True).to(DEV).eval()
    mlp.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
    means = torch.tensor(rng.standard_normal((4, 8, 3)).astype(np.float32) * 0.5, device=DEV)
    covs = torch.tensor(np.abs(rng.standard_normal((4, 8, 3))).astype(np.float32) * 1e-3, device=DEV)
    dirs = torch.nn.functional.normalize(torch.tensor(rng.standard_normal((4, 3)).astype(np.float32), device=DEV), dim=-1)
    out = mlp((means, covs), dirs)
    keys = list(out.keys())
    assert keys.index("specular_density") == keys.index("specular") + 1 and keys.index("roughness") == keys.index("specular_density") + 1
    twin = models.NerfMLP().to(DEV).eval()
    twin.load_flat_params(synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0))
    with torch.no_grad():
        twin.raw_density.weight.copy_(mlp.raw_specular_density.weight)
        twin.raw_density.bias.copy_(mlp.raw_specular_density.bias)
    twin.mark_updated() if hasattr(twin, "mark_updated") else None
    assert torch.equal(out["specular_density"], twin((means, covs), dirs)["density"])
    assert not torch.equal(out["specular_density"], out["density"])
    with pytest.raises(ValueError, match="useless"):                      # models.py:478-480
        models.NerfMLP(enable_pred_specular_density=True, use_diffuse_color=False)
    configs.clear_config()


def test_mismatched_weight_image_is_refused(hip):
    """ABI v11: the library remembers the kind of every image it packs; a level handed a pointer it packed as another
    kind returns REFNERF_EINVAL (it used to stream the wrong bytes: out-of-bounds LDS-DMA reads) -- forward and backward."""
    from refnerf_pl_amd import synthetic
    P = torch.tensor(synthetic.make_params(0, 0.05, 20.0), device=DEV)
    rays = dev_rays(synthetic.blender_rays(8, seed=4, center_frac=0.4))
    sd, w = torch.tensor([[0.0, 1.0]], device=DEV).repeat(8, 1), torch.ones((8, 1), device=DEV)
    f32_image = hip.pack_weights(P, precision=hip.PREC_F32)
    cfg = hip.default_cfg(n_samples=32, n_in=1, precision=hip.PREC_F16X2)
    with pytest.raises(ValueError, match="refnerf_level_image"):          # REFNERF_EINVAL -> ValueError, as for every bad argument
        hip.level_forward(f32_image, cfg, rays, sd, w)
    tcfg = hip.default_cfg(n_samples=32, n_in=1, precision=hip.PREC_F16X2, training=1, compute_extras=0)
    eval_image = hip.pack_weights(P, precision=hip.PREC_F16X2)
    if not hip.LEGACY_F16X2_TRAIN:
        with pytest.raises(ValueError, match="REFNERF_IMAGE_F16X2_TRAIN"):
            hip.level_forward(eval_image, tcfg, rays, sd, w, history=True, save_activations=True)
    good = hip.pack_weights(P, precision=hip.level_image(hip.PREC_F16X2, True))
    res = hip.level_forward(good, tcfg, rays, sd, w, history=True, save_activations=True)
    with pytest.raises(ValueError, match="refnerf_level_backward"):
        hip.level_backward(eval_image, tcfg, rays, res, torch.full((8, 3), 1e-2, device=DEV), None, None, torch.zeros(hip.NUM_PARAMS, device=DEV))
    # a re-pack of the same buffer as another kind is what counts from then on
    hip.level_backward(good, tcfg, rays, res, torch.full((8, 3), 1e-2, device=DEV), None, None, torch.zeros(hip.NUM_PARAMS, device=DEV))
