"""CPU: pin the oracle (oracle/refnerf_oracle.c) against vectors captured from
the upstream reference (tests/golden/make_golden.py).  SURVEY.md 8c."""
import numpy as np
import pytest

from helpers import (HIST_KEYS, MODEL_CASES, POSENC_CASES, RAYDIST_CASES, REND_KEYS, raydist_kw, TRAIN_CASES, VARIANT_CASES, posenc_params, VARIANT_HIST_KEYS, VARIANT_REND_KEYS,
                     cfg_from_bindings, load_golden, variant_params,
                     params_from_golden, rays_from_golden)
from oracle import oracle as O

WIDE_RANGE_CASES = ("model_trained_eval", "model_trained_train", "model_shiny_eval", "model_shiny_train")


def test_param_layout_matches_c():
    import ctypes as C
    from refnerf_pl_amd import layout

    class Off(C.Structure):
        _fields_ = [("sp_w", C.c_int * 8), ("sp_b", C.c_int * 8), ("sp_in", C.c_int * 8)] + \
                   [(n, C.c_int) for n in ("density_w", "density_b", "gradpred_w", "gradpred_b", "rough_w", "rough_b",
                                           "diffuse_w", "diffuse_b", "tint_w", "tint_b", "bneck_w", "bneck_b")] + \
                   [("vd_w", C.c_int * 8), ("vd_b", C.c_int * 8), ("vd_in", C.c_int * 8),
                    ("rgb_w", C.c_int), ("rgb_b", C.c_int), ("total", C.c_int)]
    o = Off()
    O.lib().rn_param_layout(C.byref(o))
    assert o.total == layout.NUM_PARAMS
    for i in range(8):
        s = layout.SPEC_BY_NAME[f"spatial_net.{i}"]
        assert (o.sp_w[i], o.sp_b[i], o.sp_in[i]) == (s.w_off, s.b_off, s.in_dim)
        s = layout.SPEC_BY_NAME[f"viewdir_mlp.{i}"]
        assert (o.vd_w[i], o.vd_b[i], o.vd_in[i]) == (s.w_off, s.b_off, s.in_dim)
    for cname, pname in (("density", "raw_density"), ("gradpred", "grad_pred"), ("rough", "raw_roughness"),
                         ("diffuse", "raw_rgb_diffuse"), ("tint", "raw_tint"), ("bneck", "bottleneck"), ("rgb", "rgb")):
        s = layout.SPEC_BY_NAME[pname]
        assert (getattr(o, cname + "_w"), getattr(o, cname + "_b")) == (s.w_off, s.b_off)


def test_sampler_indices_bit_exact_and_sdist():
    g = load_golden("sampler")
    for i in range(int(g["num_cases"])):
        t, w, n = g[f"c{i}_t"], g[f"c{i}_w"], int(g[f"c{i}_n"])
        assert np.array_equal(O.linspace_u(n), g[f"c{i}_u"]), "torch.linspace restatement"
        lg = O.resample_logits(t, w)
        ref_lg = g[f"c{i}_logits"]
        fin = np.isfinite(ref_lg)
        assert np.array_equal(np.isfinite(lg), fin)
        np.testing.assert_allclose(lg[fin], ref_lg[fin], rtol=0, atol=1e-6)
        sd, idx = O.sample_intervals(t, ref_lg, n)
        assert np.array_equal(idx, g[f"c{i}_idx"]), f"case {i}: CDF bin indices must be bit-exact"
        # sdist is a float: off = (u-cw_i)/(cw_{i+1}-cw_i) amplifies the 1-ulp
        # softmax-sum differences inside near-empty bins (cases 2,4).
        np.testing.assert_allclose(sd, g[f"c{i}_sdist"], rtol=0, atol=2e-5)
        assert np.all(np.diff(sd, axis=-1) >= 0) and sd.min() >= 0 and sd.max() <= 1
    # level 0 (single unit interval) is exactly reproducible
    for i in (0, 1, 5):
        sd, _ = O.sample_intervals(g[f"c{i}_t"], g[f"c{i}_logits"], int(g[f"c{i}_n"]))
        assert np.array_equal(sd, g[f"c{i}_sdist"])


@pytest.mark.parametrize("fam", ["blender", "llff"])
def test_cast_rays_and_ipe(fam):
    g = load_golden("cast_ipe")
    sd, td = g[fam + "_sdist"], g[fam + "_tdist"]
    mine = np.array([[O.lib().rn_s_to_t(float(s), float(nr), float(fr)) for s in row]
                     for row, nr, fr in zip(sd, g[fam + "_near"][:, 0], g[fam + "_far"][:, 0])], np.float32)
    assert np.array_equal(mine, td)
    lm, lv, mx = O.cast_samples(g[fam + "_origins"], g[fam + "_directions"], g[fam + "_radii"], td)
    assert np.array_equal(mx, g[fam + "_means"]), "sample means must be bit-exact (IPE is chaotic in them)"
    assert np.array_equal(lm, g[fam + "_lmean"])
    np.testing.assert_allclose(lv, g[fam + "_lvar"], rtol=1e-6, atol=0)
    covs = g[fam + "_covs"]
    np.testing.assert_allclose(lv, np.stack([covs[..., 2, 2], covs[..., 1, 1], covs[..., 0, 0]], -1), rtol=1e-6)
    np.testing.assert_allclose(O.ipe(g[fam + "_lmean"], g[fam + "_lvar"]), g[fam + "_ipe"], rtol=0, atol=5e-7)
    lmc, lvc, _ = O.cast_samples(g[fam + "_origins"], g[fam + "_directions"], g[fam + "_radii"], td, ray_shape=1)
    assert np.array_equal(lmc, g[fam + "_cyl_lmean"])
    np.testing.assert_allclose(lvc, g[fam + "_cyl_lvar"], rtol=1e-6)


def test_safe_sin():
    g = load_golden("cast_ipe")
    x = g["safe_sin_x"]
    f = O.ipe(np.stack([x, x, x], -1), np.zeros((len(x), 3), np.float32))
    np.testing.assert_allclose(f[:, 0], g["safe_sin_y"], rtol=0, atol=2e-7)


def test_ide_against_reference_and_fp64():
    """the reference's monomial evaluation has its own fp32 error
    (4e-3 at kappa_inv=0).  (i) the reference-order restatement reproduces the
    reference; (ii) the stable evaluation used on the path is never worse than
    the reference w.r.t. float64 truth."""
    g = load_golden("ide")
    xyz = g["xyz"]
    for kap in (0.0, 0.01, 0.1, 0.3, 1.0):
        ref = g[f"ide_{kap}"]
        truth = O.ide(xyz, kap, "f64")
        np.testing.assert_allclose(O.ide(xyz, kap, "ref32"), ref, rtol=0, atol=5e-6)
        stable = O.ide(xyz, kap, "stable")
        err_ref = np.abs(ref - truth).max()
        err_stable = np.abs(stable - truth).max()
        assert err_stable <= err_ref + 1e-5, (kap, err_stable, err_ref)
        assert err_stable < 5e-6
        if kap >= 0.1:
            np.testing.assert_allclose(stable, ref, rtol=0, atol=1e-6)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_mlp_per_sample(mode):
    g = load_golden("mlp")
    P = params_from_golden(g)
    lm, lv = g["lmean"].reshape(-1, 3), g["lvar"].reshape(-1, 3)
    v = np.repeat(g["viewdirs"], g["lmean"].shape[1], axis=0)
    res = O.mlp_samples(P, O.default_cfg(training=int(mode == "train")), lm, lv, v)
    for k in ("density", "rgb", "normals_pred", "grad_pred", "tint", "diffuse", "specular", "roughness"):
        np.testing.assert_allclose(res[k], g[f"{mode}_{k}"].reshape(res[k].shape), rtol=0, atol=2e-6, err_msg=k)
    if mode == "train":
        np.testing.assert_allclose(res["normals"], g["train_normals"].reshape(-1, 3), rtol=0, atol=5e-6)
    else:
        assert "eval_normals" not in g.files  # normals=None in eval (models.py:603)


def test_alpha_weights_and_compositing():
    g = load_golden("render")
    for op in (0, 1):
        w = O.alpha_weights(g["density"], g["tdist"], g["dirs"], bool(op))
        np.testing.assert_allclose(w, g[f"weights_opaque{op}"], rtol=0, atol=2e-7)
        assert np.all(w >= 0) and np.all(w.sum(-1) <= 1 + 1e-6)


RENDER_MODES = ("none", "linear", "norm_linear", "srgb", "norm_srgb")


def check_render_modes(render_fn):
    """render.volumetric_rendering (render.py:152-254) in all five `srgb_mapping` modes (:186-216) against the
    reference's outputs on the same per-sample inputs (tests/golden/render.npz).  `render_fn(mode, g)` -> dict with
    the r_* names of the level outputs.  Shared by the oracle (CPU) and the HIP stage entry (GPU)."""
    g = load_golden("render")
    worst = {}
    for mode in RENDER_MODES:
        res = render_fn(mode, g)
        np.testing.assert_allclose(res["weights"], g["weights_opaque0"], rtol=0, atol=2e-7)
        for k in ("rgb", "diffuse", "specular", "distance", "acc", "normals", "normals_pred", "roughness", "tint",
                  "distance_mean"):
            ref = g[f"{mode}_{k}"]
            mine = np.asarray(res["r_" + k]).reshape(ref.shape)
            # distance sums t ~ 2..6 weights: 1 ulp there is 5e-7
            tol = 2e-6 if k in ("distance", "distance_mean") else 1e-6
            np.testing.assert_allclose(mine, ref, rtol=0, atol=tol, err_msg=f"{mode} {k}")
            worst[k] = max(worst.get(k, 0.0), float(np.abs(mine - ref).max()))
        pc = np.stack([g[f"{mode}_distance_percentile_5"], g[f"{mode}_distance_median"], g[f"{mode}_distance_percentile_95"]], -1)
        assert res["r_percentiles"].dtype == np.float64
        np.testing.assert_allclose(res["r_percentiles"], pc, rtol=0, atol=1e-5, err_msg=mode)
    return worst


def test_volumetric_rendering_all_render_time_modes():
    def fn(mode, g):
        return O.render_rays(g["density"], g["tdist"], g["dirs"], g["far"], g["rgbs"], g["dif"], g["spc"], g["normals"],
                             g["normals_pred"], g["roughness"], g["tint"], render_srgb_mode=mode)
    check_render_modes(fn)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_model_end_to_end(name):
    g = load_golden(name)
    P = params_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    train = name.endswith("train")
    outs = O.model_forward(P, rays_from_golden(g), training=int(train), **lv, **kw)
    for L, res in enumerate(outs):
        assert np.array_equal(res["sdist"], g[f"L{L}_h_sdist"]) or L > 0  # level 0 sampling is exact
        for k in HIST_KEYS:
            a = g[f"L{L}_h_{k}"].reshape(res[k].shape)
            tol = 2e-4 if k == "normals_pred" else (1e-4 if k == "density" else 2e-6)
            rtol = 0
            if name in WIDE_RANGE_CASES:
                # shiny (un-attenuated degree-16 IDE terms in a K = 457 GEMM) / trained-like weights (densities ~30,
                # weights up to 0.26): summation-order noise is 3.5e-6 per sample at level 0; at level 1 the sample
                # POSITIONS already differ by an ulp (sdist ~1e-7) and the trained network
                # turns that into <= 1.4e-3 per sample (measured) -- the renderings below stay at the 1e-5 level
                wide, rtol = (5e-6, 2e-5) if L == 0 else (5e-3 if name.startswith("model_trained") else 5e-5, 2e-4)
                tol = max(tol, wide)
                if k in ("sdist", "weights"):
                    tol = 2e-6 if L == 0 else 5e-5
            np.testing.assert_allclose(res[k], a, rtol=rtol, atol=tol, err_msg=f"L{L} {k}")
        for k in REND_KEYS:
            a = g[f"L{L}_r_{k}"].reshape(res["r_" + k].shape)
            tol = 2e-5 if name in WIDE_RANGE_CASES else 5e-6
            if k == "distance_mean":
                # exp(sum(w log t) / acc): a ratio of two sums that are both ~acc -- round-off scales with 1 / acc
                # (rays that miss the trained sphere have acc ~1e-4)
                tol = tol + 1e-6 / np.maximum(res["r_acc"], 1e-6)
                assert np.all(np.abs(res["r_" + k] - a) <= tol), (L, k, np.abs(res["r_" + k] - a).max())
                continue
            np.testing.assert_allclose(res["r_" + k], a, rtol=0, atol=tol, err_msg=f"L{L} r_{k}")
        # the headline parity bar: RGB L-inf <= 1e-4 vs the reference CPU path
        assert np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max() <= 1e-4
        pc = np.stack([g[f"L{L}_r_distance_percentile_5"], g[f"L{L}_r_distance_median"],
                       g[f"L{L}_r_distance_percentile_95"]], -1)
        assert res["r_percentiles"].dtype == np.float64
        if name in WIDE_RANGE_CASES:
            # the float64 interpolation divides by the CDF step of one bin: round-off of the weights (1e-7) over a
            # near-empty bin moves the percentile inside that bin -- bulk tight, worst case bounded by the bin width
            perr = np.abs(res["r_percentiles"] - pc)
            assert np.mean(perr <= 2e-5) >= 0.95 and perr.max() <= 5e-4, (np.mean(perr <= 2e-5), perr.max())
        else:
            np.testing.assert_allclose(res["r_percentiles"], pc, rtol=0, atol=2e-5)
        if train:
            n_ref = g[f"L{L}_h_normals"].reshape(res["normals"].shape)
            err = np.abs(res["normals"] - n_ref).max(-1)
            # density-gradient normals at level 1 are ill-conditioned where the
            # gradient is tiny: gate the bulk tightly, the tail loosely.
            assert np.median(err) < 1e-4 and np.mean(err < 1e-3) > 0.97, (np.median(err), np.mean(err < 1e-3))
            np.testing.assert_allclose(res["r_normals"], g[f"L{L}_r_normals"], rtol=0, atol=2e-3)


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_training_step_losses_and_gradients(name):
    """rn_level_train (forward + data / orientation / predicted-normal losses +
    backward) against the reference's autograd: loss values and
    the parameter gradients (every 97th element + per-tensor L2 norms)."""
    from refnerf_pl_amd import layout
    g = load_golden(name)
    P = params_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    losses, grads, _ = O.model_train(P, rays_from_golden(g), g["gt_rgb"], **lv, **kw)
    # trained-like weights: the data loss is ~9e-3, i.e. fp32 round-off of the renderings (1e-7) is 1e-5 of it
    lrel = 5e-5 if name in WIDE_RANGE_CASES else 2e-6
    assert losses["data"] == pytest.approx(float(g["loss_data"]), rel=lrel)
    assert losses["orientation"] == pytest.approx(float(g["loss_orientation"]), rel=2e-4)
    assert losses["normal"] == pytest.approx(float(g["loss_normal"]), rel=2e-4)
    assert losses["total"] == pytest.approx(float(g["loss_total"]), rel=lrel)
    ref = g["grads_sub"]
    mine = grads[::97]
    # trained-like weights: the level-1 sample positions differ from the reference's by an ulp (see above) and the
    # trained network is ~5e3 x more sensitive to them than the random-init one: gradient rel-L2 1.6e-4 (measured)
    wide = name.startswith("model_trained")
    rel = np.linalg.norm(mine - ref) / np.linalg.norm(ref)
    assert rel < (5e-4 if wide else 1e-4), rel
    np.testing.assert_allclose(mine, ref, rtol=0, atol=(2e-5 if wide else 1e-6) * max(1.0, np.abs(ref).max() / 1e-3))
    norms = g["grads_tensor_l2"]
    for i, s in enumerate(layout.PARAM_SPECS):      # all 46 tensors receive their gradient (A10)
        n = s.out_dim * s.in_dim
        trel = 1e-2 if wide else 2e-3
        assert np.linalg.norm(grads[s.w_off:s.w_off + n]) == pytest.approx(norms[i, 0], rel=trel), s.name
        assert np.linalg.norm(grads[s.b_off:s.b_off + s.out_dim]) == pytest.approx(norms[i, 1], rel=trel), s.name
        assert norms[i, 0] > 0


@pytest.mark.parametrize("name", VARIANT_CASES)
def test_variant_embedding_matches_reference(name):
    """a NerfMLP with net_width_viewdirs = 128 and without n.v / tint / roughness heads, as the reference
    builds it, against the Ref-NeRF network holding the SAME weights embedded in zeros and run with roughness_bias =
    ROUGHNESS_OFF_BIAS (layout.variant_layout) -- the embedding is what the HIP path ships for these flags, so this pins
    it to the reference's own outputs, losses and autograd gradients."""
    from refnerf_pl_amd import layout
    g = load_golden(name)
    canon, _, idx = variant_params(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    kw["roughness_bias"] = layout.ROUGHNESS_OFF_BIAS
    train = name.endswith("train")
    outs = O.model_forward(canon, rays_from_golden(g), training=int(train), **lv, **kw)
    for L, res in enumerate(outs):
        assert not res["roughness"].any()                       # softplus(-1e30) = 0 exactly
        assert np.all(res["tint"] == 0.5)                       # sigmoid(0) exactly: specular = 0.5 rgb
        for k in VARIANT_HIST_KEYS:
            a = g[f"L{L}_h_{k}"].reshape(res[k].shape)
            tol = 2e-4 if k == "normals_pred" else (1e-4 if k == "density" else 2e-6)
            np.testing.assert_allclose(res[k], a, rtol=0, atol=tol, err_msg=f"L{L} {k}")
        for k in VARIANT_REND_KEYS:
            a = g[f"L{L}_r_{k}"].reshape(res["r_" + k].shape)
            if k == "distance_mean":                            # a ratio of two sums ~acc: round-off scales with 1 / acc
                assert np.all(np.abs(res["r_" + k] - a) <= 5e-6 + 1e-6 / np.maximum(res["r_acc"], 1e-6)), (L, k)
                continue
            np.testing.assert_allclose(res["r_" + k], a, rtol=0, atol=5e-6, err_msg=f"L{L} r_{k}")
        assert np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max() <= 1e-4
    if not train:
        return
    has_normals = "loss_normal" in g.files
    losses, grads, _ = O.model_train(canon, rays_from_golden(g), g["gt_rgb"], **lv, **kw,
                                     **({} if has_normals else {"normal_mults": (0.0, 0.0)}))
    assert losses["data"] == pytest.approx(float(g["loss_data"]), rel=2e-6)
    assert losses["orientation"] == pytest.approx(float(g["loss_orientation"]), rel=2e-4)
    if has_normals:
        assert losses["normal"] == pytest.approx(float(g["loss_normal"]), rel=2e-4)
    assert losses["total"] == pytest.approx(float(g["loss_total"]), rel=2e-6)
    mine = grads[idx]                                           # the variant's elements of the canonical gradient
    ref = g["grads_sub"]
    rel = np.linalg.norm(mine[::61] - ref) / np.linalg.norm(ref)
    assert rel < 1e-4, rel
    specs, _ = layout.variant_layout(net_width_viewdirs=128, use_n_dot_v=False, use_specular_tint=False, enable_pred_roughness=False)
    norms = g["grads_tensor_l2"]
    for i, s in enumerate(specs):
        n = s.out_dim * s.in_dim
        assert np.linalg.norm(mine[s.w_off:s.w_off + n]) == pytest.approx(norms[i, 0], rel=2e-3), s.name
        assert np.linalg.norm(mine[s.b_off:s.b_off + s.out_dim]) == pytest.approx(norms[i, 1], rel=2e-3), s.name
    # what the embedding adds stays inert: the dead units and the absent heads / columns receive exactly zero gradient
    # except the two head rows whose INPUT is live (raw_tint: d specular / d tint != 0; raw_roughness: sigmoid(-1e30) = 0)
    rest = np.ones(layout.NUM_PARAMS, bool)
    rest[idx] = False
    tint = layout.SPEC_BY_NAME["raw_tint"]
    rest[tint.w_off:tint.b_off + tint.out_dim] = False
    vd = [layout.SPEC_BY_NAME[f"viewdir_mlp.{i}"] for i in range(8)]
    for s in vd:                                               # dead units' OUTGOING columns see zero activations: zero
        w = grads[s.w_off:s.w_off + s.out_dim * s.in_dim].reshape(s.out_dim, s.in_dim)
        assert not w[128:, :].any(), s.name                    # dead rows: relu'(0) = 0
    assert not grads[layout.SPEC_BY_NAME["raw_roughness"].w_off:layout.SPEC_BY_NAME["raw_roughness"].b_off + 1].any()


@pytest.mark.parametrize("name", RAYDIST_CASES)
def test_raydist_and_disable_integration_match_reference(name):
    """Model.raydist_fn (coord.construct_ray_warps, coord.py:63-99) and Model.disable_integration (models.py:228-231):
    the oracle's s_to_t per function / zero covariances against the reference's outputs; the training fixture also pins
    losses and gradients with both switched on.  Without the integration the degree-15 features sin(2^15 x) are not
    attenuated: at level 1, where the sample positions differ from the reference's by an ulp (sdist ~7e-7, DESIGN.md
    section 2), the network answers with up to 1.3e-2 in a density -- level 0 stays at round-off (1e-7), renderings at 1e-4."""
    g = load_golden(name)
    P = params_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    kw.update(raydist_kw(g))
    train = name.endswith("train")
    noint = bool(int(g["disable_integration"]))
    outs = O.model_forward(P, rays_from_golden(g), training=int(train), **lv, **kw)
    for L, res in enumerate(outs):
        for k in HIST_KEYS:
            a = g[f"L{L}_h_{k}"].reshape(res[k].shape)
            tol = 2e-4 if k == "normals_pred" else (1e-4 if k == "density" else 2e-6)
            if L > 0 and k not in ("sdist", "weights"):
                tol = max(tol, 5e-5)
            if L > 0 and noint:
                tol = 5e-5 if k == "sdist" else (2e-4 if k == "weights" else (3e-2 if k in ("density", "normals_pred") else 1e-3))
            np.testing.assert_allclose(res[k], a, rtol=0, atol=tol, err_msg=f"L{L} {k}")
        for k in REND_KEYS:
            a = g[f"L{L}_r_{k}"].reshape(res["r_" + k].shape)
            base = 2e-4 if (noint and L > 0) else 5e-6
            if k in ("distance_mean", "distance"):
                # metric distances: the warped functions put samples at t up to `far` with fp32 round-off of the warp
                assert np.all(np.abs(res["r_" + k] - a) <= 4 * base + 1e-6 / np.maximum(res["r_acc"], 1e-6)), (L, k, np.abs(res["r_" + k] - a).max())
                continue
            np.testing.assert_allclose(res["r_" + k], a, rtol=0, atol=base, err_msg=f"L{L} r_{k}")
        assert np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max() <= 1e-4
    if train:
        losses, grads, _ = O.model_train(P, rays_from_golden(g), g["gt_rgb"], **lv, **kw)
        assert losses["data"] == pytest.approx(float(g["loss_data"]), rel=2e-4)
        assert losses["orientation"] == pytest.approx(float(g["loss_orientation"]), rel=5e-3)
        assert losses["normal"] == pytest.approx(float(g["loss_normal"]), rel=5e-3)
        ref = g["grads_sub"]
        rel = np.linalg.norm(grads[::97] - ref) / np.linalg.norm(ref)
        print(name, "gradient rel-L2 vs reference", rel)
        assert rel < 2e-2, rel


@pytest.mark.parametrize("name", POSENC_CASES)
def test_posenc_view_encoding_matches_reference(name):
    """`use_directional_enc = False` (coord.pos_enc of the reflected direction, internal/models.py:487-492, coord.py:136-147):
    the oracle's ide_mode = 2 writes the 33 features into the directional slots, the reference's [256, 162] / [256, 418]
    weights are embedded at those slots -- against the reference's outputs, losses and autograd gradients."""
    from refnerf_pl_amd import layout
    g = load_golden(name)
    canon, _, idx = posenc_params(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    kw["ide_mode"] = 2
    train = name.endswith("train")
    outs = O.model_forward(canon, rays_from_golden(g), training=int(train), **lv, **kw)
    for L, res in enumerate(outs):
        for k in HIST_KEYS:
            a = g[f"L{L}_h_{k}"].reshape(res[k].shape)
            tol = 2e-4 if k == "normals_pred" else (1e-4 if k == "density" else 2e-6)
            np.testing.assert_allclose(res[k], a, rtol=0, atol=tol, err_msg=f"L{L} {k}")
        for k in REND_KEYS:
            a = g[f"L{L}_r_{k}"].reshape(res["r_" + k].shape)
            if k == "distance_mean":
                assert np.all(np.abs(res["r_" + k] - a) <= 5e-6 + 1e-6 / np.maximum(res["r_acc"], 1e-6)), (L, k)
                continue
            np.testing.assert_allclose(res["r_" + k], a, rtol=0, atol=5e-6, err_msg=f"L{L} r_{k}")
        assert np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max() <= 1e-4
    if not train:
        return
    losses, grads, _ = O.model_train(canon, rays_from_golden(g), g["gt_rgb"], **lv, **kw)
    assert losses["data"] == pytest.approx(float(g["loss_data"]), rel=2e-6)
    assert losses["orientation"] == pytest.approx(float(g["loss_orientation"]), rel=2e-4)
    # (the density-gradient normals of this fixture sit a little further out in round-off than the others: 3.1e-4)
    assert losses["normal"] == pytest.approx(float(g["loss_normal"]), rel=6e-4)
    mine = grads[idx]
    ref = g["grads_sub"]
    assert np.linalg.norm(mine[::61] - ref) / np.linalg.norm(ref) < 1e-4
    specs, _ = layout.variant_layout(use_directional_enc=False, deg_view=5)
    norms = g["grads_tensor_l2"]
    for i, s in enumerate(specs):
        n = s.out_dim * s.in_dim
        assert np.linalg.norm(mine[s.w_off:s.w_off + n]) == pytest.approx(norms[i, 0], rel=2e-3, abs=1e-12), s.name
    assert norms[[s.name for s in specs].index("raw_roughness"), 0] == 0      # pos_enc ignores the roughness


@pytest.mark.parametrize("tag,spec", [("blender", (800, 800, 1111.111, None)), ("llff", (1008, 756, 815.0, 1.0))])
def test_ray_generator_matches_reference(tag, spec):
    """The numpy ray generator behind every synthetic fixture / bench batch restates
    camera_utils.pixels_to_rays (+ convert_to_ndc): pinned by vectors from the reference."""
    from refnerf_pl_amd import synthetic
    g = load_golden("camera")
    w, h, focal, ndc = spec
    res = synthetic._pixels_to_rays(g[tag + "_pix_x"].astype(np.int64), g[tag + "_pix_y"].astype(np.int64), focal, w, h,
                                    g[tag + "_camtoworld"].astype(np.float64), ndc_near=ndc)
    for k, a in zip(("origins", "directions", "viewdirs", "radii", "imageplane"), res):
        np.testing.assert_allclose(a.reshape(g[f"{tag}_{k}"].shape), g[f"{tag}_{k}"], rtol=0, atol=3e-7, err_msg=k)


def test_dilation_and_anneal_vs_reference():
    """Model.dilation_* (models.py:167-186; host stepfun.max_dilate_weights) and Model.anneal_slope
    (models.py:188-201) at train_frac 0.3: the dilated, annealed resampling of level 1 and its rendering
    against the reference (fixture model_dilation_anneal_eval)."""
    import torch
    from refnerf_pl_amd import stepfun
    g = load_golden("model_dilation_anneal_eval")
    P = params_from_golden(g)
    rays = rays_from_golden(g)
    R = rays["origins"].shape[0]
    anneal = (10. * 0.3) / ((10. - 1) * 0.3 + 1)
    sd0 = np.tile(np.array([[0.0, 1.0]], np.float32), (R, 1))
    lvl0 = O.level_forward(P, O.default_cfg(n_samples=64, n_in=1, anneal=anneal), rays, sd0, np.ones((R, 1), np.float32))
    np.testing.assert_array_equal(lvl0["sdist"], g["L0_h_sdist"])
    np.testing.assert_allclose(lvl0["weights"], g["L0_h_weights"], atol=2e-7)
    dilation = 0.0025 + 0.5 * (1.0 - 0.0) / 64
    t, w = stepfun.max_dilate_weights(torch.tensor(g["L0_h_sdist"]), torch.tensor(g["L0_h_weights"]), dilation,
                                      domain=(0.0, 1.0), renormalize=True)
    t, w = t[..., 1:-1].numpy(), w[..., 1:-1].numpy()
    assert t.shape == (R, 3 * 64 - 1) and w.shape == (R, 3 * 64 - 2)
    lvl1 = O.level_forward(P, O.default_cfg(n_samples=64, n_in=w.shape[1], anneal=anneal), rays, t, w)
    # annealed logits (0.81 * log(w + padding)) go through exp again: the oracle's exp (rn_det_expf) and torch's differ
    # by an ulp, which moves interpolated knots by a few 1e-7 -- not the bin they fall in
    assert np.mean(np.abs(lvl1["sdist"] - g["L1_h_sdist"]) < 2e-6) > 0.999
    ok = np.abs(lvl1["sdist"] - g["L1_h_sdist"]).max(-1) < 2e-6
    np.testing.assert_allclose(lvl1["r_rgb"][ok], g["L1_r_rgb"][ok], atol=5e-6)
    np.testing.assert_allclose(lvl1["weights"][ok], g["L1_h_weights"][ok], atol=5e-6)


@pytest.mark.parametrize("fam", ["blender", "llff_linear"])
def test_unfused_torch_path_matches_oracle(fam):
    """oracle/torch_path.py (the unfused PyTorch-CPU restatement bench.py times as the second cpu_baseline) against the
    C oracle in its reference-order IDE mode: identical CDF bin indices, renderings to 5e-6."""
    from oracle import torch_path as T
    from refnerf_pl_amd import synthetic
    P = synthetic.make_params(0, 0.05, 20.0)
    if fam == "blender":
        rays, okw, tkw = synthetic.blender_rays(40, seed=1, center_frac=0.5), {}, {}
    else:
        rays = synthetic.llff_rays(24, seed=1)
        okw = dict(srgb_mapping=0, render_srgb_mode="norm_linear")
        tkw = dict(srgb_mapping=False, render_srgb_mode="norm_linear")
    a = O.model_forward(P, rays, num_prop_samples=64, num_nerf_samples=96, ide_mode=1, **okw)
    b = T.model_forward(P, rays, num_prop_samples=64, num_nerf_samples=96, **tkw)
    for L in range(2):
        assert np.array_equal(a[L]["bin_idx"], b[L]["bin_idx"])
        for k in ("sdist", "weights", "rgb", "r_rgb", "r_diffuse", "r_specular", "r_acc", "r_distance", "r_distance_mean",
                  "r_roughness", "r_tint", "r_normals_pred", "r_percentiles"):
            np.testing.assert_allclose(b[L][k].reshape(a[L][k].shape), a[L][k], rtol=2e-6, atol=5e-6, err_msg=f"L{L} {k}")


@pytest.mark.parametrize("tag", ["trained_long", "trained_llff"])
def test_oracle_on_the_long_trained_weights(tag):
    """the harsher trained-like fixture (2500 reference steps at lr 1e-3, fp32 blob; tests/golden/make_golden.py::
    golden_trained_long) and the forward-facing one (1200 steps on NDC rays, linear colour, norm_linear render map: the C4 / C5
    family): the oracle against the reference's eval outputs -- level-0 samples bit for bit, rendered RGB 2e-5
    (the reference's BLAS summation order vs the oracle's sequential fp32 sums on activations up to 136)"""
    import os
    from helpers import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, f"model_{tag}_eval.npz")):
        pytest.skip("fixture not generated")
    g = load_golden(f"model_{tag}_eval")
    P = params_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    out = O.model_forward(P, rays_from_golden(g), **lv, **kw)
    assert np.array_equal(out[0]["sdist"], g["L0_h_sdist"].reshape(out[0]["sdist"].shape))
    for L in range(2):
        err = float(np.abs(out[L]["r_rgb"] - g[f"L{L}_r_rgb"]).max())
        print(f"L{L}: oracle RGB L-inf vs reference {err:.2e}")
        assert err <= 2e-5


# ---------------------------------------------------------------- round 6: the float64 build and the shared logit log
def test_det_logf_is_the_correctly_rounded_fp32_log():
    """include/refnerf_detmath.h::rn_det_logf (the log of the resampling logits, models.py:200-203, shared by the kernels
    and the oracle) against float64 log rounded to fp32: identical on 2 M values over the whole positive range -- dense around 1,
    around the resample padding 0.01 and at the subnormal boundary --, and its special cases."""
    import ctypes as C
    lib = O.lib()
    rng = np.random.default_rng(0)
    bits = np.concatenate([rng.integers(1, 0x7f800000, 1_000_000, dtype=np.int64),
                           np.float32(1.0).view(np.int32) + rng.integers(-400000, 400000, 400_000),
                           np.float32(0.01).view(np.int32) + rng.integers(0, 1 << 24, 400_000),
                           rng.integers(1, 0x01000000, 200_000, dtype=np.int64)]).astype(np.int32)
    x = bits.view(np.float32)
    t = np.zeros(x.size + 1, np.float32)
    t[1:] = 1.0                                             # rn_resample_logits: logits[i] = anneal * rn_det_logf(w[i] + padding) where t[i + 1] > t[i]
    t = np.cumsum(t).astype(np.float32)
    out = np.empty(x.size, np.float32)
    FP = C.POINTER(C.c_float)
    lib.rn_resample_logits(t.ctypes.data_as(FP), x.ctypes.data_as(FP), C.c_int(x.size), C.c_float(1.0), C.c_float(0.0), out.ctypes.data_as(FP))
    want = np.log(x.astype(np.float64)).astype(np.float32)
    assert np.array_equal(out, want), int((out != want).sum())
    sp = np.array([0.0, -1.0, np.inf, np.nan, 1.0], np.float32)
    o5 = np.empty(5, np.float32)
    lib.rn_resample_logits(t[:6].ctypes.data_as(FP), sp.ctypes.data_as(FP), C.c_int(5), C.c_float(1.0), C.c_float(0.0), o5.ctypes.data_as(FP))
    assert o5[0] == -np.inf and np.isnan(o5[1]) and o5[2] == np.inf and np.isnan(o5[3]) and o5[4] == 0.0


@pytest.mark.parametrize("name", [n for n in MODEL_CASES if n.endswith("eval")] + ["model_trained_long_eval", "model_trained_llff_eval"])
def test_f64_build_is_within_fp32_rounding_of_the_reference(name):
    """oracle/oracle_f64.py (the SAME restatement compiled with double for float, `make -C oracle librefnerf_oracle_f64.so`)
    against the reference's own fp32 outputs on every eval fixture: the float64 build is the function the reference evaluates,
    so the two differ by the reference's fp32 rounding only -- 1e-6 on random-init networks, up to 1e-4 on rays of the trained
    ones whose level-1 samples are ill-conditioned (that distance IS what the GPU tests' float64 gate measures); level-0
    sample positions (no MLP in front of them) agree to an fp32 ulp."""
    import os
    from helpers import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, name + ".npz")):
        pytest.skip(name + ".npz missing")
    from oracle import oracle_f64 as O64
    g = load_golden(name)
    P, rays = params_from_golden(g), rays_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    truth = O64.model_forward(P, rays, **lv, **kw)
    mine = O.model_forward(P, rays, **lv, **kw)
    trained = "trained" in name
    for L, res in enumerate(truth):
        assert res["r_rgb"].dtype == np.float64
        d_ref = float(np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max())
        d_o32 = float(np.abs(res["r_rgb"] - mine[L]["r_rgb"]).max())
        assert d_ref <= (2e-4 if trained else 5e-6), (name, L, d_ref)
        # the fp32 oracle IS the reference's arithmetic (pinned to 1e-6; 1e-5 on the trained sets' conditioned rays)
        assert abs(d_ref - d_o32) <= (1.5e-5 if trained else 2e-6), (name, L, d_ref, d_o32)
        if L == 0:
            np.testing.assert_allclose(res["sdist"], g["L0_h_sdist"].reshape(res["sdist"].shape), rtol=0, atol=1.2e-7)
        assert np.abs(res["weights"] - g[f"L{L}_h_weights"].reshape(res["weights"].shape)).max() <= (5e-4 if trained else 2e-5)
