"""GPU: REFNERF_PREC_F16X2 -- the parity-grade 16-bit mode (split-operand f16 MFMA, include/refnerf_hip.h) -- against
the CPU oracle and the reference's golden vectors.  The bar is north_star's: rendered RGB L-inf <= 1e-4, also on the
trained-like weights where the plain bf16 / f16 modes measure 5e-2 / 1e-2 (tests/test_hip_parity.py), with >= 99.9 %
identical CDF bin indices.  Matches the reference's fp32 nn.Linear arithmetic (internal/models.py:576-580, 686-700)."""
import numpy as np
import pytest

from helpers import (EVAL_CASES, cfg_from_bindings, load_golden, params_from_golden, rays_from_golden, trained_blob)
from test_hip_parity import DEV, O, _psnr, _record, hip, run_hip_model  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu

F16X2 = 3
RGB_TOL = 1e-4          # north_star: "outputs match the reference CPU path on identical rays within 1e-4 RGB L-inf"


def perturbed_trained_blob():
    """The trained-like blob is stored as float16, i.e. its weights have NO low part; a real checkpoint is fp32.  This
    variant multiplies every weight by (1 + 1e-3 N(0,1)) in fp32, so that the W_lo fragments carry information (without
    the weight split this blob measures 2e-3: scripts/exp_split_precision.py)."""
    rng = np.random.default_rng(5)
    b = trained_blob()
    return (b * (1.0 + 1e-3 * rng.standard_normal(b.shape))).astype(np.float32)


@pytest.mark.parametrize("name", EVAL_CASES)
def test_f16x2_vs_reference_fixtures(hip, name):
    """every eval fixture captured from the reference (random-init, sharpened, LLFF, C1, shiny, trained-like):
    rendered RGB within 1e-4 (measured ~1e-6 .. 1e-5) and level-0 bin indices identical to the reference's."""
    g = load_golden(name)
    P = params_from_golden(g)
    rays = rays_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    outs = run_hip_model(hip, P, rays, kw, lv, precision=F16X2)
    f32 = run_hip_model(hip, P, rays, kw, lv, precision=0)
    rec = {}
    for L, res in enumerate(outs):
        rec[f"L{L}_rgb_linf_vs_reference"] = float(np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max())
        rec[f"L{L}_f32_mode_rgb_linf_vs_reference"] = float(np.abs(f32[L]["r_rgb"] - g[f"L{L}_r_rgb"]).max())
        rec[f"L{L}_weights_linf_vs_reference"] = float(np.abs(res["weights"] - g[f"L{L}_h_weights"]).max())
        rec[f"L{L}_bin_idx_vs_f32_mode"] = float(np.mean(res["bin_idx"] == f32[L]["bin_idx"]))
        assert np.all(np.isfinite(res["r_percentiles"]))
    _record("f16x2_fixture_" + name, rec)
    assert np.array_equal(outs[0]["sdist"], g["L0_h_sdist"].reshape(outs[0]["sdist"].shape))   # level 0: the reference's samples, bit for bit
    for L in range(len(outs)):
        assert rec[f"L{L}_rgb_linf_vs_reference"] <= (2e-5 if "trained" not in name else RGB_TOL), rec
        assert rec[f"L{L}_weights_linf_vs_reference"] <= 1e-4, rec
        assert rec[f"L{L}_bin_idx_vs_f32_mode"] >= 0.999, rec       # (fixtures of 16-64 rays: one tie is already 1e-4 of them)


def _tail(err_per_ray):
    """max / 99.99th percentile / count over north_star's 1e-4 of a per-ray RGB L-inf error"""
    return float(err_per_ray.max()), float(np.quantile(err_per_ray, 0.9999)), int((err_per_ray > RGB_TOL).sum())


INDEX_FLOOR = 0.9999     # end to end >= 99.99 % identical CDF bin indices (measured >= 99.999 %: the differing ones are CDF
                         # ties one ulp apart; the sampler STAGE is bit-exact: test_sampler_*); r03 asserted 0.999
# ---- the float64 gate (round 6; the same construction SURVEY H2 uses for the IDE) ----
# north_star's bar compares two fp32-grade evaluations of an ill-conditioned function: on rays that graze a thin surface the
# REFERENCE's own fp32 arithmetic is 6e-5 .. 1.3e-4 away from its exact value (oracle/oracle_f64.py: the same restatement in
# float64; measured on the 2500-step weights, profiles/r06/parity_full_size.json), because fp32 rounding of the sample
# coordinates and of the coarse weights moves the fine samples.  On every ray
#     |hip - f64| <= max(1e-4, K * |fp32 oracle - f64|)
# i.e. a ray may pass 1e-4 only where the reference's own rounding error is of that size, and then by at most the factor K:
#   K = 4 for the f16x2 mode: its spatial operands carry 22 significand bits against fp32's 24 -- unit roundoff 2^(24-22) = 4 x;
#   K = 2 for the f32 mode: the same unit roundoff in another summation order (MFMA accumulation vs the k-ordered fma chain).
# Measured (7 trained-weights batches, 38,912 rays): 2 rays of the f16x2 mode pass 1e-4 against float64 (1.12e-4 and 1.15e-4,
# ratios 3.9 and 1.9), 1 ray of the f32 mode (1.31e-4, ratio 1.04), 1 ray of the fp32 oracle itself (1.26e-4).  Fifteen further
# seeded batches (scripts/parity_f64_sweep.py, 92,160 rays): 0 violations; rays over 1e-4 against float64: f16x2 5, f32 mode 4,
# the fp32 oracle 5; largest ratio on such a ray 1.5.
K_F16X2, K_F32 = 4.0, 2.0


@pytest.fixture(scope="module")
def O64():
    from oracle import oracle_f64
    return oracle_f64


def f64_gate(hip_rgb, o32_rgb, o64_rgb, k):
    """-> (violations, max |hip - f64|, max |o32 - f64|, rays of hip over 1e-4 vs f64, largest ratio on those rays)"""
    e = np.abs(hip_rgb - o64_rgb).max(-1)
    er = np.abs(o32_rgb - o64_rgb).max(-1)
    over = e > RGB_TOL
    ratio = float((e[over] / np.maximum(er[over], 1e-30)).max()) if over.any() else 0.0
    return int((e > np.maximum(RGB_TOL, k * er)).sum()), float(e.max()), float(er.max()), int(over.sum()), ratio


@pytest.mark.parametrize("case", ["C2_trained_like", "C2_trained_like_fp32_weights", "C2_bench_batch", "C3_shiny", "C2_trained_long",
                                  "C3_trained_long", "C4_trained_llff",
                                  # round 5 (one seeded batch per weight set is thin evidence for a 4 % margin):
                                  # two more views of the 2500-step weights at C3's shape, the LLFF weights at C5's sample count
                                  "C3_trained_long_view2", "C3_trained_long_view3", "C5_trained_llff"])
def test_f16x2_full_size_vs_oracle(hip, O, O64, case):
    """BASELINE-sized batches on the HIP path against the CPU oracle: >= 99.99 % identical bin indices at every level and
    rendered RGB within north_star's 1e-4 of the fp32 oracle -- except on rays where the float64 build of the same oracle shows
    the reference's own fp32 rounding error to be of that size (the float64 gate above, asserted on EVERY ray in both parity
    modes; the plain 1e-4 against the fp32 oracle on every ray of every batch on which the oracle itself stays within 5e-5 of
    float64; maxima and 99.99th percentiles against the fp32 oracle recorded for all).  Trained-like weights (f16-exact as stored, and
    perturbed to full fp32 precision), the bench batch (C2), the shiny network (C3, the ring-of-records kernel variant); the
    harsher weight sets (trained_long: 2500 reference steps, three views; trained_llff: the forward-facing family of C4 / C5)
    go through both oracles as WHOLE batches.  The fine level ALONE, fed the fp32 oracle's coarse step function, must
    reproduce the oracle's bin indices and sdist BIT FOR BIT (shared rn_det_logf / rn_det_expf) and its RGB to 1e-4."""
    from refnerf_pl_amd import synthetic
    R, N, n_or = (8192, 192, 512) if case.startswith("C3") else ((2048, 256, 2048) if case.startswith("C5") else (4096, 128, 512))
    kw = {}
    llff_seed = 3
    if "trained_long" in case:                  # the 2500-step fp32 weight set, also through the ring-of-records variant (C3 shape)
        from helpers import trained_long_blob
        view = {"": 3, "_view2": 11, "_view3": 23}[case.split("trained_long")[1]]
        P, rk, n_or = trained_long_blob(), dict(seed=view, center_frac=0.8), R
    elif case.endswith("trained_llff"):         # forward-facing NDC rays, linear colour + norm_linear render map (llff_refnerf.gin)
        from helpers import trained_llff_blob
        g = load_golden("model_trained_llff_eval")
        kw = cfg_from_bindings(g["bindings"])[0]
        P, rk, n_or = trained_llff_blob(), None, R
        llff_seed = 7 if case.startswith("C5") else 3
    elif case == "C2_trained_like":
        P, rk = trained_blob(), dict(seed=3, center_frac=0.8)
    elif case == "C2_trained_like_fp32_weights":
        P, rk = perturbed_trained_blob(), dict(seed=3, center_frac=0.8)
    elif case == "C2_bench_batch":
        P, rk = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0), dict(seed=1, center_frac=0.5)
    else:
        P, rk = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0, roughness_bias=-6.0), dict(seed=1, center_frac=0.5)
    rays = synthetic.llff_rays(R, seed=llff_seed) if rk is None else synthetic.blender_rays(R, **rk)
    lv = dict(num_prop_samples=N, num_nerf_samples=N)
    out = run_hip_model(hip, P, rays, kw, lv, precision=F16X2)
    again = run_hip_model(hip, P, rays, kw, lv, precision=F16X2)
    f32 = run_hip_model(hip, P, rays, kw, lv, precision=0)
    ref = O.model_forward(P, {k: v[:n_or] for k, v in rays.items()}, **lv, **kw)
    truth = O64.model_forward(P, {k: v[:n_or] for k, v in rays.items()}, history=False, **lv, **kw)
    rec = {"rays": R, "samples": N, "oracle_rays": n_or}
    for L in range(2):
        a = out[L]
        for k in a:
            assert np.array_equal(a[k], again[L][k]), f"non-deterministic {k}"
        w, sd = a["weights"], a["sdist"]
        assert np.isfinite(a["r_rgb"]).all()
        assert w.min() >= 0 and np.all(w.sum(-1) <= 1 + 1e-5)
        assert np.all(np.diff(sd, axis=-1) >= 0) and sd.min() >= 0 and sd.max() <= 1
        if not kw:
            bg = np.maximum(0, 1 - a["r_acc"])[:, None]
            np.testing.assert_allclose(a["r_rgb"], (w[..., None] * a["rgb"]).sum(1) + bg, rtol=0, atol=3e-5)
        e16 = np.abs(a["r_rgb"][:n_or] - ref[L]["r_rgb"]).max(-1)
        e32 = np.abs(f32[L]["r_rgb"][:n_or] - ref[L]["r_rgb"]).max(-1)
        rec[f"L{L}_rgb_linf_vs_oracle"], rec[f"L{L}_rgb_p9999_vs_oracle"], rec[f"L{L}_rays_over_1e-4"] = _tail(e16)
        rec[f"L{L}_f32_mode_rgb_linf_vs_oracle"], rec[f"L{L}_f32_mode_rgb_p9999_vs_oracle"], rec[f"L{L}_f32_mode_rays_over_1e-4"] = _tail(e32)
        rec[f"L{L}_weights_linf_vs_oracle"] = float(np.abs(a["weights"][:n_or] - ref[L]["weights"]).max())
        same = a["bin_idx"][:n_or] == ref[L]["bin_idx"]
        rec[f"L{L}_bin_idx_agreement"] = float(np.mean(same))
        rec[f"L{L}_bin_idx_differing"] = f"{int(same.size - same.sum())} of {same.size}"
        same32 = f32[L]["bin_idx"][:n_or] == ref[L]["bin_idx"]
        rec[f"L{L}_f32_mode_bin_idx_differing"] = f"{int(same32.size - same32.sum())} of {same32.size}"
        rec[f"L{L}_sdist_max_abs_diff"] = float(np.abs(a["sdist"][:n_or] - ref[L]["sdist"]).max())
        rec[f"L{L}_psnr_vs_oracle_db"] = _psnr(a["r_rgb"][:n_or], ref[L]["r_rgb"])
        rec[f"L{L}_rgb_linf_vs_f32_mode_full_batch"] = float(np.abs(a["r_rgb"] - f32[L]["r_rgb"]).max())
        rec[f"L{L}_rgb_p9999_vs_f32_mode_full_batch"] = float(np.quantile(np.abs(a["r_rgb"] - f32[L]["r_rgb"]).max(-1), 0.9999))
        rec[f"L{L}_density_max"] = float(a["density"].max())
        # float64 columns: |hip - f64| per mode, |fp32 oracle - f64|, and the gate
        for tag, res, k in (("", a, K_F16X2), ("f32_mode_", f32[L], K_F32)):
            bad, emax, ermax, n_over, ratio = f64_gate(res["r_rgb"][:n_or], ref[L]["r_rgb"], truth[L]["r_rgb"], k)
            rec[f"L{L}_{tag}rgb_linf_vs_f64"] = emax
            rec[f"L{L}_{tag}rays_over_1e-4_vs_f64"] = n_over
            rec[f"L{L}_{tag}worst_ratio_to_the_oracles_own_f64_error_on_those_rays"] = ratio
            rec[f"L{L}_{tag}f64_gate_violations"] = bad
        rec[f"L{L}_oracle_f32_rgb_linf_vs_f64"] = float(np.abs(ref[L]["r_rgb"] - truth[L]["r_rgb"]).max())
        rec[f"L{L}_oracle_f32_rays_over_1e-4_vs_f64"] = int((np.abs(ref[L]["r_rgb"] - truth[L]["r_rgb"]).max(-1) > RGB_TOL).sum())
        rec[f"L{L}_oracle_f32_weights_linf_vs_f64"] = float(np.abs(ref[L]["weights"] - truth[L]["weights"]).max())
    # the fine level ALONE: fed the oracle's own step function (sdist / weights of ITS coarse level), so that the level kernel's
    # arithmetic separates from the resampler's conditioning (a coarse weight 4e-7 off moves a fine sample by 6e-6 on rays that
    # graze a thin surface, and the colour by 30 x that: scripts/dbg_worst_ray.py)
    import torch
    from test_hip_parity import dev_rays
    sub = dev_rays({k: v[:n_or] for k, v in rays.items()})
    for prec, tag in ((F16X2, ""), (0, "f32_mode_")):
        packed = hip.pack_weights(torch.tensor(P, device=DEV), precision=hip.level_image(prec, False, 0))
        cfg1 = hip.default_cfg(n_samples=N, n_in=N, precision=prec, **kw)
        res = hip.level_forward(packed, cfg1, sub, torch.tensor(ref[0]["sdist"], device=DEV), torch.tensor(ref[0]["weights"], device=DEV))
        es = np.abs(res["r_rgb"].cpu().numpy() - ref[1]["r_rgb"]).max(-1)
        rec[f"L1_{tag}rgb_linf_given_the_oracles_step_function"] = float(es.max())
        rec[f"L1_{tag}bin_idx_agreement_given_the_oracles_step_function"] = float(np.mean(res["bin_idx"].cpu().numpy() == ref[1]["bin_idx"]))
        rec[f"L1_{tag}sdist_bit_equal_given_the_oracles_step_function"] = bool(np.array_equal(res["sdist"].cpu().numpy(), ref[1]["sdist"]))
        if not tag:      # ... and on the ray that is worst end to end: what of ITS error is the fine level's own
            worst = int(np.abs(out[1]["r_rgb"][:n_or] - ref[1]["r_rgb"]).max(-1).argmax())
            rec["L1_worst_ray"] = worst
            rec["L1_worst_ray_rgb_err_given_the_oracles_step_function"] = float(es[worst])
    print(case, rec)
    _record("f16x2_" + case, rec)
    for tag in ("", "f32_mode_"):
        # identical (sdist, weights) in: the fused level's resampler is index- and position-exact (kernel and oracle share the
        # logit's log and the softmax's exp, include/refnerf_detmath.h), its colour within the bar
        assert rec[f"L1_{tag}bin_idx_agreement_given_the_oracles_step_function"] == 1.0, rec
        assert rec[f"L1_{tag}sdist_bit_equal_given_the_oracles_step_function"], rec
        assert rec[f"L1_{tag}rgb_linf_given_the_oracles_step_function"] <= RGB_TOL, rec
    for L in range(2):
        assert rec[f"L{L}_f64_gate_violations"] == 0 and rec[f"L{L}_f32_mode_f64_gate_violations"] == 0, rec
        # (against the fp32 oracle the two modes -- the f32 mode as often as f16x2 -- pass 1e-4 on single rays of the sharp Blender
        #  batches, up to 1.4e-4 on twelve further views, 99.99th percentile up to 1.2e-4: profiles/r06/parity_f64_sweep.json,
        #  92,160 rays, 0 gate violations.  Those figures are recorded above; what is ASSERTED against the fp32 oracle is the plain
        #  bar wherever the oracle itself is converged:)
        if rec[f"L{L}_oracle_f32_rgb_linf_vs_f64"] <= 0.5 * RGB_TOL:
            # the reference's own rounding error is small on this batch: the plain bar holds on every ray, in both modes and between them
            assert rec[f"L{L}_rgb_linf_vs_oracle"] <= RGB_TOL and rec[f"L{L}_f32_mode_rgb_linf_vs_oracle"] <= RGB_TOL, rec
            assert rec[f"L{L}_rgb_linf_vs_f32_mode_full_batch"] <= RGB_TOL, rec
        assert rec[f"L{L}_bin_idx_agreement"] >= INDEX_FLOOR, rec
    assert rec["L0_bin_idx_agreement"] == 1.0, rec        # level 0 does not depend on the MLP: bit-exact resampler


@pytest.mark.parametrize("R,n0,n1", [(3, 64, 64), (5, 192, 256), (2, 33, 2), (37, 40, 72), (2051, 192, 192), (4099, 96, 96), (1, 128, 128)])
def test_f16x2_ragged_shapes_vs_f32_mode(hip, R, n0, n1):
    """ragged / partly filled passes (idle waves, 16-sample runs past the end, the ring-of-records variant) against the
    f32 parity mode on the trained-like weights"""
    from refnerf_pl_amd import synthetic
    P = perturbed_trained_blob()
    rr = synthetic.blender_rays(R, seed=R, center_frac=0.6)
    lv = dict(num_prop_samples=n0, num_nerf_samples=n1)
    x = run_hip_model(hip, P, rr, {}, lv, precision=F16X2)
    y = run_hip_model(hip, P, rr, {}, lv, precision=0)
    for L in range(2):
        assert np.array_equal(x[0]["bin_idx"], y[0]["bin_idx"])
        assert np.abs(x[L]["r_rgb"] - y[L]["r_rgb"]).max() <= RGB_TOL, (R, n0, n1, L)
        assert np.mean(x[L]["bin_idx"] == y[L]["bin_idx"]) >= 0.995
        assert np.abs(x[L]["weights"] - y[L]["weights"]).max() <= 2e-4


@pytest.mark.parametrize("kw", [dict(opaque_background=1), dict(ray_shape=1), dict(srgb_mapping=0, render_srgb_mode=2),
                                dict(disable_integration=1), dict(raydist=2), dict(dir_enc=1)],
                         ids=["opaque", "cylinder", "linear_norm", "nointegration", "reciprocal", "posenc"])
def test_f16x2_level_options(hip, kw):
    """the level-cfg switches take the same code paths as the other 16-bit kernels: check them against the f32 mode"""
    from refnerf_pl_amd import synthetic
    P = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0)
    rr = synthetic.blender_rays(96, seed=11, center_frac=0.5)
    lv = dict(num_prop_samples=64, num_nerf_samples=96)
    x = run_hip_model(hip, P, rr, kw, lv, precision=F16X2)
    y = run_hip_model(hip, P, rr, kw, lv, precision=0)
    # zero covariances (nointegration) leave the degree-15 IPE features unattenuated: layer 0 then sums 96 O(1) terms with
    # heavy cancellation, where 22 significand bits (this mode) and 24 (the f32 mode) differ visibly in the small composites
    # (normals_pred of near-empty rays); the RGB bar stays north_star's
    loose = bool(kw.get("disable_integration"))
    for L in range(2):
        assert np.abs(x[L]["r_rgb"] - y[L]["r_rgb"]).max() <= (RGB_TOL if loose else 2e-5), (kw, L)
        for k in ("r_diffuse", "r_specular", "r_acc", "r_distance_mean", "r_normals_pred", "r_roughness", "r_tint"):
            np.testing.assert_allclose(x[L][k], y[L][k], rtol=0, atol=5e-4 if loose else 5e-5, err_msg=k)
        # (distances of 2 .. 6: a percentile moves by the CDF knot's width times the relative weight difference)
        np.testing.assert_allclose(x[L]["r_percentiles"], y[L]["r_percentiles"], rtol=0, atol=1e-3 if loose else 1e-4)


def test_f16x2_through_the_model_api(hip):
    """Config.hip_precision = 'f16x2' through Model.__call__ and render_image's chunk loop; training levels refuse it"""
    import os
    import torch
    from refnerf_pl_amd import configs, models, utils, synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(root, "configs", "refnerf_blender.gin")],
                                            ["Config.hip_precision = 'f16x2'", "Model.num_prop_samples = 64", "Model.num_nerf_samples = 64"])
    cfg = configs.Config()
    model = models.construct_model(None, cfg).to(DEV).eval()
    blob = perturbed_trained_blob()
    model.nerf_mlp.load_flat_params(blob)
    rays_np = synthetic.blender_rays(200, seed=4, center_frac=0.7)
    rays = utils.rays_from_dict(dict(rays_np), torch.device(DEV))
    with torch.no_grad():
        a = model(rays, 1.0, True)
        cfg.hip_precision = "f32"
        b = model(rays, 1.0, True)
    for L in range(2):
        assert float((a[0][L]["rgb"] - b[0][L]["rgb"]).abs().max()) <= RGB_TOL
        assert a[0][L]["distance_median"].dtype == torch.float64
    cfg.hip_train_precision = "f16"          # 'f16' is an inference mode ('f16x2' is a training mode too: see below)
    model.train()
    with pytest.raises(ValueError):
        model(rays, 1.0, True)
    # the split-f16 training forward saves split-f16 pair units (REFNERF_ACT_F16X2): the backward must be the split-f16 one
    cfg.hip_train_precision, cfg.hip_bwd_precision = "f16x2", "f32"
    with pytest.raises(ValueError, match="hip_bwd_precision"):
        model(rays, 1.0, True)
    cfg.hip_train_precision = cfg.hip_bwd_precision = "f32"


# ---------------------------------------------------------------- split-f16 chains in the training forward
@pytest.mark.parametrize("wgrad", ["bf16x3", "f16"])       # 22-bit GEMM inputs / one half ('f16': what the shipped configs select)
@pytest.mark.parametrize("bwd", ["f32", "f16x2"])
@pytest.mark.parametrize("name", ["model_blender_sharp_train", "model_llff_linear_train", "model_shiny_train", "model_trained_train"])
def test_f16x2_chain_training_step_vs_reference(hip, O, name, bwd, wgrad):
    """Config.hip_train_precision = hip_bwd_precision = 'f16x2': the split-f16 training kernels (round 5: REFNERF_ACT_SQ
    activations, two-product backward, f16 weight-gradient GEMM) against the REFERENCE's own losses and autograd gradients on the
    golden training fixtures, the trained-like one included.  Same bars as the exact-fp32 chains (test_training_step_gradients):
    gradient rel-L2 2e-4 (1e-3 trained-like), loss 1e-5, rendered RGB 1e-4 -- where the bf16 chains measure 1e-2 / 1e-1.
    `bwd` selects the backward of the exact-fp32 FORWARD that runs beside it: the f32 chains, or the split chains on fp32 rows
    (level_bwd_f16x2c_r32, 22-bit deltas) -- that leg is asserted against the same bar, and the two modes against each other."""
    import os
    import torch
    from refnerf_pl_amd import configs, layout, models, train_utils, utils
    g = load_golden(name)
    bindings = [str(b) for b in g["bindings"] if str(b)]
    res = {}
    for mode in ("f16x2", "f32"):
        # the split-f16 forward saves split-f16 pair units, which only the split-f16 backward reads; `bwd` selects the backward
        # of the exact-fp32 forward beside it (f32 chains, or the split chains on fp32 rows: level_bwd_f16x2c_r32)
        configs.clear_config()
        configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                                bindings + [f"Config.hip_train_precision = '{mode}'",
                                                            f"Config.hip_bwd_precision = '{'f16x2' if mode == 'f16x2' else bwd}'",
                                                            f"Config.hip_wgrad_mode = '{wgrad}'"])
        cfg = configs.Config()
        model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).train()
        model.nerf_mlp.load_flat_params(params_from_golden(g))
        rays = utils.rays_from_dict(rays_from_golden(g), DEV)
        batch = utils.Batch(rays=rays, rgb=np.asarray(g["gt_rgb"], np.float32))
        rend, hist = model(rays, 1.0, False)
        total, terms, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
        total.backward()
        grads = torch.cat([p.grad.flatten() for p in model.nerf_mlp.ordered_parameters()]).cpu().numpy()
        res[mode] = (grads, float(total.detach()), rend[1]["rgb"].detach().cpu().numpy(), hist[1]["normals"].detach().cpu().numpy(),
                     [hist[L]["normals"].detach().cpu().numpy() for L in range(2)])
    grads, total, rgb_l1, normals, normals_per_level = res["f16x2"]
    # row a10 in the mode of record (models.py:603-609): the density-gradient normals of the split-f16 training forward, per
    # sample, against the oracle and against the reference's own ray_history -- the statistics of
    # test_training_forward_density_normals (the normal is ill-conditioned where the density gradient is tiny: bulk tight, tail loose)
    kw_o, lv_o = cfg_from_bindings(g["bindings"])
    orc = O.model_forward(params_from_golden(g), rays_from_golden(g), training=1, **lv_o, **kw_o)
    nstat = {}
    for L in range(2):
        for tag, refn, bulk, tail in (("oracle", orc[L]["normals"], 2e-5, 0.99), ("reference", g[f"L{L}_h_normals"], 1e-4, 0.97)):
            err = np.abs(normals_per_level[L] - refn.reshape(normals_per_level[L].shape)).max(-1)
            nstat[f"L{L}_normals_median_err_vs_{tag}"] = float(np.median(err))
            nstat[f"L{L}_normals_frac_under_1e-3_vs_{tag}"] = float(np.mean(err < 1e-3))
            assert np.median(err) < bulk and np.mean(err < 1e-3) > tail, (name, L, tag, nstat)
    ref = g["grads_sub"]
    rel = float(np.linalg.norm(grads[::97] - ref) / np.linalg.norm(ref))
    rel32 = float(np.linalg.norm(res["f32"][0][::97] - ref) / np.linalg.norm(ref))
    rel_modes = float(np.linalg.norm(grads - res["f32"][0]) / np.linalg.norm(res["f32"][0]))
    lrel = abs(total - float(g["loss_total"])) / abs(float(g["loss_total"]))
    rgb = float(np.abs(rgb_l1 - g["L1_r_rgb"]).max())
    tn = g["grads_tensor_l2"]
    worst = max(abs(np.linalg.norm(grads[s.w_off:s.w_off + s.out_dim * s.in_dim]) / tn[i, 0] - 1.0) for i, s in enumerate(layout.PARAM_SPECS))
    nerr = float(np.abs(normals - res["f32"][3]).max())
    print(f"{name} f16x2 chains vs reference: gradient rel-L2 {rel:.2e} (f32 chains {rel32:.2e}; between the modes {rel_modes:.2e}), "
          f"worst tensor-norm error {worst:.2e}, loss rel {lrel:.2e}, RGB L-inf {rgb:.2e}, density normals vs f32 chains {nerr:.2e}")
    _record(f"f16x2_chain_training_vs_reference/{name}/bwd={bwd}/wgrad={wgrad}", dict(grad_rel_l2=rel, grad_rel_l2_f32_chains=rel32, grad_rel_l2_between_modes=rel_modes,
                                                              worst_tensor_norm_err=float(worst), loss_rel=lrel, rgb_linf=rgb, **nstat))
    trained = name.startswith("model_trained")
    assert rel < (1e-3 if trained else 2e-4), rel
    assert rel32 < (1e-3 if trained else 2e-4), (bwd, rel32)       # the f32 forward with the f32 / split-on-fp32-rows backward
    assert rel_modes < 5e-4, (bwd, rel_modes)                      # measured 5e-5 .. 2.1e-4 (11-bit deltas)
    assert lrel < 1e-5 and rgb < 1e-4
    configs.clear_config()


# ---------------------------------------------------------------- the harsher trained-like weights (2500 reference steps, fp32 blob)
def _have_long():
    import os
    from helpers import GOLDEN
    return os.path.exists(os.path.join(GOLDEN, "model_trained_long_eval.npz"))


def _have(tag):
    import os
    from helpers import GOLDEN
    return os.path.exists(os.path.join(GOLDEN, f"model_{tag}_eval.npz"))


LONG_SETS = [t for t in ("trained_long", "trained_llff") if _have(t)]      # Blender rays / forward-facing NDC rays (C4, C5)


@pytest.mark.parametrize("tag", LONG_SETS)
def test_trained_long_eval_every_mode_vs_reference(hip, O, tag):
    """"a real network will be harsher on 16-bit operands".  Weights after 2500 of the reference's own Adam
    steps (lr 1e-3, fp32 blob): the parity-grade modes hold north_star's 1e-4 RGB against the reference's outputs; the
    plain 16-bit modes are recorded."""
    g = load_golden(f"model_{tag}_eval")
    P = params_from_golden(g)
    rays = rays_from_golden(g)
    kw, lv = cfg_from_bindings(g["bindings"])
    ref = O.model_forward(P, rays, **lv, **kw)
    rec = {"max_abs_weight": float(np.abs(P).max()), "max_density": float(g["L1_h_density"].max())}
    for prec, tag in ((0, "f32"), (F16X2, "f16x2"), (1, "bf16"), (2, "f16")):
        outs = run_hip_model(hip, P, rays, kw, lv, precision=prec)
        for L, res in enumerate(outs):
            rec[f"{tag}_L{L}_rgb_linf_vs_reference"] = float(np.abs(res["r_rgb"] - g[f"L{L}_r_rgb"]).max())
            rec[f"{tag}_L{L}_bin_idx_vs_oracle"] = float(np.mean(res["bin_idx"] == ref[L]["bin_idx"]))
    rec["oracle_L1_rgb_linf_vs_reference"] = float(np.abs(ref[1]["r_rgb"] - g["L1_r_rgb"]).max())
    print(rec)
    _record(tag + "_eval", rec)
    for tag in ("f32", "f16x2"):
        for L in range(2):
            assert rec[f"{tag}_L{L}_rgb_linf_vs_reference"] <= RGB_TOL, (tag, L, rec)
            assert rec[f"{tag}_L{L}_bin_idx_vs_oracle"] >= 0.999, (tag, L, rec)      # (32-ray fixtures: one tie = 2.4e-4)


@pytest.mark.parametrize("tag", LONG_SETS)
@pytest.mark.parametrize("chains", ["f32", "f16x2"])
def test_trained_long_training_step_vs_reference(hip, chains, tag):
    """one training step on the same weights: losses and autograd gradients of the reference, exact-fp32 and split-f16 chains.
    On this sharp, trained surface the level-1 gradient is ill-conditioned in the SAMPLE POSITIONS: the 4e-7 differences of the
    level-0 weights between two arithmetics move some level-1 positions by up to 5e-6 in s (f32 mode vs the reference: 137
    of 1552 edges by > 1e-6), and un-damped IPE degrees turn that into 1e-3 of the level-1 gradient (scripts/dbg_trained_long_*.py;
    level 0 alone: 3e-6).  Hence 5e-3 here, and the isolation test below for the arithmetic itself."""
    import os
    import torch
    from refnerf_pl_amd import configs, models, train_utils, utils
    g = load_golden(f"model_{tag}_train")
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                            [str(b) for b in g["bindings"] if str(b)] +
                                            [f"Config.hip_train_precision = '{chains}'", f"Config.hip_bwd_precision = '{chains}'"])
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
    print(f"{tag} [{chains} chains]: gradient rel-L2 vs reference {rel:.2e}, loss rel {lrel:.2e}, RGB L-inf {rgb:.2e}")
    _record(tag + "_train/" + chains, dict(grad_rel_l2=rel, loss_rel=lrel, rgb_linf=rgb))
    assert rel < 5e-3 and lrel < 1e-5 and rgb < RGB_TOL
    configs.clear_config()


@pytest.mark.parametrize("tag", LONG_SETS)
def test_trained_long_split_chains_equal_f32_chains_from_the_same_step_function(hip, tag):
    """the arithmetic itself on the harsher weights: both levels run from IDENTICAL step functions in the exact-fp32 and the
    split-f16 chain mode (forward and backward), same upstream gradients -> the 1.11 M gradients agree to 7e-5 .. 2.7e-4 (11-bit
    deltas since round 4; with round 3's 22-bit deltas: 2e-6 / 2e-5), the rendered RGB to 3e-7 / 7e-6"""
    import torch
    g = load_golden(f"model_{tag}_train")
    P = torch.tensor(params_from_golden(g), device=DEV)
    rays = {k: torch.tensor(v, device=DEV) for k, v in rays_from_golden(g).items()}
    for k in ("radii", "near", "far"):
        rays[k] = rays[k].reshape(-1)
    R = rays["origins"].shape[0]
    packed = {prec: hip.pack_weights(P, precision=hip.level_image(prec, True)) for prec in (0, F16X2)}
    gen = torch.Generator().manual_seed(3)
    sd, w = torch.tensor([[0.0, 1.0]], device=DEV).repeat(R, 1), torch.ones((R, 1), device=DEV)
    for N in (64, 96):
        g_rgb = (torch.randn((R, 3), generator=gen) * 1e-2).to(DEV)
        g_w = (torch.randn((R, N), generator=gen) * 1e-3).to(DEV)
        # (no seed on normals_pred: -normalize(grad_pred) amplifies 1e-6 differences by 1 / |grad_pred| wherever the trained
        #  network predicts no gradient -- conditioning of that output, not arithmetic)
        g_np = None
        grads, outs = {}, {}
        for prec in (0, F16X2):
            cfg = hip.default_cfg(n_samples=N, n_in=w.shape[1], training=1, compute_extras=0, **cfg_from_bindings(g["bindings"])[0])
            cfg.precision = prec
            res = hip.level_forward(packed[prec], cfg, rays, sd, w, history=True, save_activations=True)
            out = torch.zeros(hip.NUM_PARAMS, device=DEV)
            hip.level_backward(packed[prec], cfg, rays, res, g_rgb, g_w, g_np, out)
            grads[prec], outs[prec] = out.cpu().numpy(), res
        assert torch.equal(outs[0]["sdist"], outs[F16X2]["sdist"])
        rel = float(np.linalg.norm(grads[F16X2] - grads[0]) / np.linalg.norm(grads[0]))
        drgb = float((outs[0]["r_rgb"] - outs[F16X2]["r_rgb"]).abs().max())
        print(f"N = {N}, n_in = {w.shape[1]}: gradient rel diff between the chain modes {rel:.2e}, rendered RGB diff {drgb:.2e}")
        # round 4: the split-f16 backward hands its layer deltas to the weight-gradient GEMM as ONE half each (11 bits: 2^-12
        # relative per element; 4e-5 .. 2e-4 of the gradient under these white-noise upstream gradients, at any sample count --
        # test_split_chain_gradients_at_full_size); with 22-bit deltas (round 3) the chain modes agreed to 2e-6 / 2e-5 here
        # round 5: the training forward runs on the eval kernel's skeleton: its directional trunk takes x as ONE half
        # ([W_hi | W_lo] x, two products): rendered RGB within 7e-6 of the f32 chains on the forward-facing set (the eval kernel,
        # whose trunk is plain f16: 2e-5), 3e-7 on the Blender sets; the backward chains carry ONE half per delta: 2e-4 / 7e-5
        assert rel < 4e-4 and drgb < 2e-5
        sd, w = outs[0]["sdist"].contiguous(), outs[0]["weights"].contiguous()      # the next level's input: the f32 step function


@pytest.mark.parametrize("wgrad", ["bf16x3", "f16"])
def test_split_chain_gradients_at_full_size(hip, wgrad):
    """(`wgrad`: the weight-gradient GEMM's spatial layer inputs at 22 bits, or -- 'f16', what the shipped configs select -- at one
    half: measured 2.0e-4 / 1.6e-4 vs 2.3e-4 / 1.7e-4 with the two-product backward of round 5.)  What the 11-bit layer deltas of the split-f16 backward (refnerf_layout.h: one half per element, half the DELTA bytes)
    cost at BASELINE size: a C2-sized level (4096 rays x 128 samples, 2500-step weights, both levels from the exact-fp32 step
    function), split-f16 chains vs exact-fp32 chains under the SAME upstream gradients.  The upstream gradients here are
    white noise, so the weight gradient is itself an incoherent sum and the 2^-12 rounding of its delta operand does not
    average away: 1.3e-4 .. 1.5e-4 of the gradient norm at 5e5 samples, the same as on the 2-3 k samples of the golden
    fixtures (with 22-bit deltas, round 3: 2e-6).  Against the reference's autograd on real losses the fixtures measure
    5e-5 .. 1.1e-4 (test_f16x2_chain_training_step_vs_reference).  Bounded here so that a regression cannot hide."""
    import torch
    from refnerf_pl_amd import synthetic
    from helpers import trained_long_blob
    R, N = 4096, 128
    P = torch.tensor(trained_long_blob(), device=DEV)
    rd = synthetic.blender_rays(R, seed=3, center_frac=0.8)
    rays = {k: torch.tensor(v, device=DEV) for k, v in rd.items()}
    for k in ("radii", "near", "far"):
        rays[k] = rays[k].reshape(-1)
    packed = {prec: hip.pack_weights(P, precision=hip.level_image(prec, True)) for prec in (0, F16X2)}
    gen = torch.Generator().manual_seed(5)
    sd, w = torch.tensor([[0.0, 1.0]], device=DEV).repeat(R, 1), torch.ones((R, 1), device=DEV)
    rec = {}
    for lvl in range(2):
        g_rgb = (torch.randn((R, 3), generator=gen) * 1e-3).to(DEV)
        g_w = (torch.randn((R, N), generator=gen) * 1e-4).to(DEV)
        grads, outs = {}, {}
        for prec in (0, F16X2):
            cfg = hip.default_cfg(n_samples=N, n_in=w.shape[1], training=1, compute_extras=0, precision=prec)
            if prec == F16X2 and wgrad == "f16":
                cfg.wgrad_mode = hip.WGRAD_F16
            res = hip.level_forward(packed[prec], cfg, rays, sd, w, history=True, save_activations=True)
            out = torch.zeros(hip.NUM_PARAMS, device=DEV)
            hip.level_backward(packed[prec], cfg, rays, res, g_rgb, g_w, None, out)
            grads[prec], outs[prec] = out.cpu().numpy(), {k: res[k] for k in ("sdist", "weights")}
            del res
        rel = float(np.linalg.norm(grads[F16X2] - grads[0]) / np.linalg.norm(grads[0]))
        rec[f"L{lvl}_grad_rel_l2_between_chain_modes"] = rel
        sd, w = outs[0]["sdist"].contiguous(), outs[0]["weights"].contiguous()
    print(wgrad, rec)
    _record("split_chain_gradients_full_size/" + wgrad, rec)
    assert all(v < 3e-4 for v in rec.values()), rec


def test_f16x2_operand_range(hip):
    """include/refnerf_hip.h: the hi halves are IEEE halves.  Inside the range (hidden activations up to ~8e3 here, a trained
    network's reach ~1e2) the mode holds its parity; beyond 65504 a unit becomes hi = inf, lo = -inf, the next layer's
    accumulators NaN, and the inference kernel's NaN-propagating ReLU carries that to the outputs: NaN, not a finite wrong
    colour."""
    from refnerf_pl_amd import layout, synthetic
    rays = synthetic.blender_rays(64, seed=2, center_frac=0.6)
    lv = dict(num_prop_samples=64, num_nerf_samples=64)
    dev = {}
    for scale in (30.0, 300.0, 3000.0):
        P = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=1.0)
        for name in ("spatial_net.2", "spatial_net.3"):                    # two layers: hidden activations ~ 0.09 scale^2
            sp = layout.SPEC_BY_NAME[name]
            P[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim] *= scale
        nxt = layout.SPEC_BY_NAME["spatial_net.4"]
        P[nxt.w_off:nxt.w_off + nxt.out_dim * nxt.in_dim] /= scale * scale   # ... and back to O(1) for the rest of the network
        f32 = run_hip_model(hip, P, rays, {}, lv, precision=0)
        x2 = run_hip_model(hip, P, rays, {}, lv, precision=F16X2)
        assert np.isfinite(f32[1]["r_rgb"]).all()
        bad = ~np.isfinite(x2[1]["r_rgb"]).all(-1)
        err = np.abs(x2[1]["r_rgb"] - f32[1]["r_rgb"]).max(-1)
        dev[scale] = (float(bad.mean()), float(np.max(np.where(bad, 0.0, err))))
        print(f"scale {scale:g} (activations up to ~{0.093 * scale * scale:.3g}): rays with NaN output {dev[scale][0]:.2f}, "
              f"worst finite deviation from the f32 mode {dev[scale][1]:.2e}")
    _record("f16x2_operand_range", {str(k): list(v) for k, v in dev.items()})
    assert dev[30.0] == (0.0, dev[30.0][1]) and dev[30.0][1] <= 5e-6 and dev[300.0][0] == 0.0 and dev[300.0][1] <= 5e-6   # in range: parity
    assert dev[3000.0][0] > 0.5 and dev[3000.0][1] <= 1e-4      # out of range (8e5): NaN, and whatever stays finite is right
    # the split training chains (also the F16X2 path of a general basis) are loud in the same way
    import torch
    packed = hip.pack_weights(torch.tensor(P, device=DEV), precision=hip.level_image(F16X2, True))
    r = {k: torch.tensor(v, device=DEV) for k, v in rays.items()}
    for k in ("radii", "near", "far"):
        r[k] = r[k].reshape(-1)
    cfg = hip.default_cfg(n_samples=64, n_in=1, training=1, compute_extras=0)
    cfg.precision = F16X2
    res = hip.level_forward(packed, cfg, r, torch.tensor([[0.0, 1.0]], device=DEV).repeat(64, 1), torch.ones((64, 1), device=DEV),
                            history=True, save_activations=True)
    assert float(torch.isnan(res["r_rgb"]).float().mean()) > 0.5


@pytest.mark.parametrize("chains,fused,wgrad", [("f32", False, "bf16x3"), ("f16x2", False, "bf16x3"), ("f16x2", True, "bf16x3"),
                                                ("f16x2", False, "f16"), ("f16x2", True, "f16")])
def test_twenty_optimiser_steps_follow_the_reference(hip, chains, fused, wgrad):
    """the whole loop, not single steps: the first 20 Adam steps of the REFERENCE from the seeded init on fixed batches
    (tests/golden/trajectory.npz) against Model + torch.optim.Adam here -- per-step losses and the accumulated parameter
    update.  Exercises the weight re-pack after every step, the gradient path into the parameters (per-tensor and the flat
    blob with the fused optimiser) and the optimiser coupling; exact-fp32 and split-f16 chains."""
    import os
    import torch
    from refnerf_pl_amd import configs, models, synthetic, train_utils, utils
    g = load_golden("trajectory")
    steps, n_rays, n_samples, lr, eps, seed = g["recipe"]
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")], [
        f"Model.num_prop_samples = {int(n_samples)}", f"Model.num_nerf_samples = {int(n_samples)}",
        f"Config.hip_train_precision = '{chains}'", f"Config.hip_bwd_precision = '{chains}'", f"Config.hip_wgrad_mode = '{wgrad}'"] +
        (["Config.hip_flat_grads = True"] if fused else []))
    cfg = configs.Config()
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).train()
    init = synthetic.make_params(seed=int(seed), bias_scale=0.0)
    model.nerf_mlp.load_flat_params(init)
    params = [model.nerf_mlp.flat_parameter()] if fused else list(model.parameters())
    opt = torch.optim.Adam(params, lr=float(lr), eps=float(eps), fused=fused)
    worst = 0.0
    for it in range(int(steps)):
        rd = synthetic.blender_rays(int(n_rays), seed=9100 + it, center_frac=0.85)
        rays = utils.rays_from_dict(rd, DEV)
        batch = utils.Batch(rays=rays, rgb=g["gt_rgb"][it])
        opt.zero_grad(set_to_none=True)
        rend, hist = model(rays, 1.0, False)
        total, terms, _ = train_utils.compute_losses(model, batch, rays, rend, hist, cfg)
        total.backward()
        opt.step()
        model.nerf_mlp.mark_updated()
        rel = abs(float(total.detach()) - g["loss_total"][it]) / g["loss_total"][it]
        worst = max(worst, rel)
        assert rel < 2e-5, (it, float(total.detach()), g["loss_total"][it])          # measured 7e-7 over all 20 steps
        assert float(terms["data"].detach()) == pytest.approx(g["loss_data"][it], rel=2e-5)
    upd = (model.nerf_mlp.flat_params().detach().cpu().numpy() - init)[::97]
    rel_u = float(np.linalg.norm(upd - g["update_sub"]) / np.linalg.norm(g["update_sub"]))
    print(f"[{chains} chains, fused={fused}] worst per-step loss deviation {worst:.2e}; accumulated update vs the reference's: rel-L2 {rel_u:.2e}")
    _record(f"trajectory/{chains}/fused={fused}/wgrad={wgrad}", dict(worst_loss_rel=worst, update_rel_l2=rel_u))
    # measured 7e-4 (f32 chains) / 4.6e-3 (split-f16 chains of round 5: two-product backward, 11-bit deltas; round 4: 1.3e-3) --
    # Adam's 1 / sqrt(v) on tiny gradients.  Bars at ~2x the measured values, so that further drift is caught
    assert rel_u < (1.5e-3 if chains == "f32" else 9e-3), rel_u
    configs.clear_config()


@pytest.mark.skipif(not _have("trained_long"), reason="tests/golden/model_trained_long_eval.npz missing")
def test_render_from_a_reference_format_checkpoint(hip, tmp_path):
    """SURVEY 8f-3 end to end on the GPU: the weights the reference reached after 2500 Adam steps, written as a Lightning
    checkpoint with the reference's key names (`model.nerf_mlp.spatial_net.0.weight` ... -- nerf_system.py:22-33), loaded
    through utils.load_reference_checkpoint into a Model on the device and rendered through the HIP path: the renderings
    equal the ones the REFERENCE computed from those weights (tests/golden/model_trained_long_eval.npz)."""
    import os
    import torch
    from refnerf_pl_amd import configs, models, utils
    from helpers import trained_long_blob
    g = load_golden("model_trained_long_eval")
    gin = os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")
    bindings = [str(b) for b in g["bindings"] if str(b)]
    configs.clear_config()
    configs.parse_config_files_and_bindings([gin], bindings)
    writer = models.construct_model(utils.dummy_rays(), configs.Config())            # (host side only: a CPU module)
    writer.nerf_mlp.load_flat_params(trained_long_blob())
    path = tmp_path / "last.ckpt"
    torch.save(utils.reference_checkpoint(writer, epoch=0, global_step=2500), path)
    keys = sorted(torch.load(path, weights_only=False)["state_dict"])
    assert "model.nerf_mlp.spatial_net.0.weight" in keys and "model.prop_mlp.rgb.bias" in keys and len(keys) == 92
    del writer

    configs.clear_config()
    configs.parse_config_files_and_bindings([gin], bindings)
    cfg = configs.Config()
    torch.manual_seed(1)
    model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).eval()           # random init: every weight must come from the file
    missing, unexpected = utils.load_reference_checkpoint(model, str(path))
    assert missing == [] and unexpected == []
    assert np.array_equal(model.nerf_mlp.flat_params().detach().cpu().numpy(), trained_long_blob())
    rays = utils.rays_from_dict(rays_from_golden(g), DEV)
    rec = {}
    for prec in ("f16x2", "f32"):
        cfg.hip_precision = prec
        with torch.no_grad():
            rend, hist = model(rays, 1.0, True)
        for L in range(2):
            rec[f"{prec}_L{L}_rgb_linf_vs_reference"] = float(np.abs(rend[L]["rgb"].cpu().numpy() - g[f"L{L}_r_rgb"]).max())
            rec[f"{prec}_L{L}_acc_linf_vs_reference"] = float(np.abs(rend[L]["acc"].cpu().numpy() - g[f"L{L}_r_acc"]).max())
    print(rec)
    _record("reference_checkpoint_render", rec)
    assert all(v <= RGB_TOL for v in rec.values()), rec
    configs.clear_config()


@pytest.mark.skipif(not _have("trained_long"), reason="tests/golden/trained_long_blob.npz missing")
def test_f16x2_image_crop_vs_oracle(hip, O):
    """the tail of the mode of record against the REFERENCE ARITHMETIC over an image region, not against
    the f32 mode.  A 200 x 200 crop of an 800 x 800 Blender-style view of the 2500-step weights -- rows 300..499, columns
    560..759: the interior of the sphere, its silhouette (grazing rays: the worst-conditioned ones) and background -- cast on
    the device, rendered through models.render_image (10 chunks of 4096 rays) in the f16x2 and the f32 mode, every one of
    the 40,000 rays through the CPU oracle.  Recorded: max, 99.99th percentile, rays over 1e-4; asserted: north_star's bar."""
    import os
    import functools
    import torch
    from refnerf_pl_amd import camera_utils, configs, models, synthetic, utils
    from helpers import trained_long_blob
    configs.clear_config()
    configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")], [])
    cfg = configs.Config()
    model = models.construct_model(None, cfg).to(DEV).eval()
    blob = trained_long_blob()
    model.nerf_mlp.load_flat_params(blob)
    c2w, focal = synthetic.blender_camera(seed=1)
    full = camera_utils.cast_pinhole_rays(c2w.astype(np.float32), 800, 800, focal, 2.0, 6.0, device=torch.device(DEV))
    y0, x0, n = 300, 560, 200
    crop = utils.Rays(**{k: getattr(full, k)[y0:y0 + n, x0:x0 + n].contiguous() for k in full.__dataclass_fields__})
    rays_np = {k: getattr(crop, k).reshape(n * n, -1).cpu().numpy() for k in ("origins", "directions", "viewdirs", "radii", "near", "far", "lossmult")}
    ref = O.model_forward(blob, rays_np, num_prop_samples=128, num_nerf_samples=128, history=False)
    want = ref[-1]["r_rgb"].reshape(n, n, 3)
    rec = {"pixels": n * n, "acc_range": [float(ref[-1]["r_acc"].min()), float(ref[-1]["r_acc"].max())]}
    err = {}
    for prec in ("f16x2", "f32"):
        cfg.hip_precision = prec
        with torch.no_grad():
            img = models.render_image(functools.partial(model, train_frac=1.0, compute_extras=True), crop, cfg, verbose=False, device=torch.device(DEV))
        e = np.abs(img["rgb"].cpu().numpy() - want).max(-1)
        err[prec] = e
        rec[prec + "_rgb_linf_vs_oracle"], rec[prec + "_rgb_p9999_vs_oracle"], rec[prec + "_pixels_over_1e-4"] = _tail(e.reshape(-1))
        rec[prec + "_psnr_vs_oracle_db"] = _psnr(img["rgb"].cpu().numpy(), want)
    worst = np.unravel_index(np.argmax(err["f16x2"]), err["f16x2"].shape)
    rec["worst_pixel"] = [int(worst[0]) + y0, int(worst[1]) + x0]
    rec["f32_mode_at_worst_f16x2_pixel"] = float(err["f32"][worst])
    rec["acc_at_worst_pixel"] = float(ref[-1]["r_acc"].reshape(n, n)[worst])
    print(rec)
    _record("f16x2_image_crop_vs_oracle", rec)
    assert 0.0 <= rec["acc_range"][0] < 0.05 and rec["acc_range"][1] > 0.95      # the crop holds background AND surface
    assert rec["f16x2_rgb_linf_vs_oracle"] <= RGB_TOL and rec["f32_rgb_linf_vs_oracle"] <= RGB_TOL, rec
    configs.clear_config()


def test_split_f16_activation_format_contract(hip):
    """ABI v10: refnerf_activations_format(cfg) names what refnerf_level_forward_train writes -- REFNERF_ACT_SQ = 3 (pair units /
    one-half rows / lane-local sign words of the round-5 kernels) for the split-f16 chains on the built-in basis, fp32 rows for a
    general basis, bf16 pair-rows for the bf16 chains -- and refnerf_level_backward refuses the one combination nothing serves
    (those activations with the exact-fp32 or bf16 chains) instead of reading them as fp32 rows."""
    import ctypes as C
    import torch
    from refnerf_pl_amd import synthetic
    fmt = lambda **kw: int(hip.lib().refnerf_activations_format(C.byref(hip.default_cfg(n_samples=32, n_in=1, training=1, **kw))))
    assert fmt(precision=0) == 0 and fmt(precision=1) == 1 and fmt(precision=F16X2) == (2 if hip.LEGACY_F16X2_TRAIN else 3)
    assert fmt(precision=F16X2, ipe_groups=7) == 0 and fmt(precision=0, ipe_groups=2) == 0
    P = torch.tensor(synthetic.make_params(0, 0.05, 20.0), device=DEV)
    packed = hip.pack_weights(P, precision=hip.level_image(F16X2, True))
    rd = synthetic.blender_rays(8, seed=4, center_frac=0.4)
    rays = {k: torch.tensor(v, device=DEV) for k, v in rd.items()}
    for k in ("radii", "near", "far"):
        rays[k] = rays[k].reshape(-1)
    R, N = 8, 32
    sd, w = torch.tensor([[0.0, 1.0]], device=DEV).repeat(R, 1), torch.ones((R, 1), device=DEV)
    cfg = hip.default_cfg(n_samples=N, n_in=1, training=1, compute_extras=0, precision=F16X2)
    res = hip.level_forward(packed, cfg, rays, sd, w, history=True, save_activations=True)
    assert res["activations_format"] == (2 if hip.LEGACY_F16X2_TRAIN else 3)
    g_rgb = torch.full((R, 3), 1e-2, device=DEV)
    out = torch.zeros(hip.NUM_PARAMS, device=DEV)
    hip.level_backward(packed, cfg, rays, res, g_rgb, None, None, out)            # the pairing that exists
    assert torch.isfinite(out).all() and float(out.abs().max()) > 0
    for bad_prec in (0, 1):
        bad = type(cfg).from_buffer_copy(cfg)
        bad.precision = bad_prec
        with pytest.raises(hip.HipLibraryError, match="REFNERF_ACT_F16X2|REFNERF_ACT_SQ|split-f16"):
            hip.level_backward(packed, bad, rays, res, g_rgb, None, None, torch.zeros(hip.NUM_PARAMS, device=DEV))


@pytest.mark.shipped_config
def test_shipped_config_runs_the_mode_of_record(hip):
    """a drop-in user who loads configs/refnerf_blender.gin WITHOUT bindings gets the mode of record -- the
    split-f16 kernels in inference, training forward and backward -- not the strict-parity f32 kernels: the three hip_* knobs of
    the shipped config, the precision Model hands the library, the kernel that then runs (bit-identical output to an explicit
    'f16x2' binding, different from 'f32'), the activation format of a training level, and the timer family of its launches."""
    import os
    import torch
    from refnerf_pl_amd import configs, models, synthetic, utils
    gin = os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")
    lv = ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 64"]
    rays = utils.rays_from_dict(dict(synthetic.blender_rays(64, seed=1, center_frac=0.4)), DEV)
    blob = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=20.0)
    out = {}
    for tag, extra in (("shipped", []), ("f16x2", ["Config.hip_precision = 'f16x2'"]), ("f32", ["Config.hip_precision = 'f32'"])):
        configs.clear_config()
        configs.parse_config_files_and_bindings([gin], lv + extra)
        cfg = configs.Config()
        if tag == "shipped":
            assert (cfg.hip_precision, cfg.hip_train_precision, cfg.hip_bwd_precision, cfg.hip_wgrad_mode) == ("f16x2", "f16x2", "f16x2", "f16")
        model = models.construct_model(None, cfg).to(DEV).eval()
        model.nerf_mlp.load_flat_params(blob)
        if tag == "shipped":
            assert model._level_cfg(model.nerf_mlp, 64, 1, 1.0, True).precision == hip.PREC_F16X2
        with torch.no_grad():
            rend, _ = model(rays, 1.0, True)
        out[tag] = rend[1]["rgb"].clone()
        if tag == "shipped":
            # a training level of the shipped config: the round-5 kernels (their activation format), one forward + one backward launch
            model.train()
            hip.set_timing(True)
            rend, hist = model(rays, 1.0, True)
            (rend[0]["rgb"].sum() + rend[1]["rgb"].sum()).backward()
            torch.cuda.synchronize()
            fams = {f: hip.get_timing(f)[1] for f in (hip.TIMER_FORWARD, hip.TIMER_BACKWARD, hip.TIMER_WGRAD)}
            hip.set_timing(False)
            assert fams[hip.TIMER_FORWARD] == 2 and fams[hip.TIMER_BACKWARD] == 2 and fams[hip.TIMER_WGRAD] == 2, fams
            tcfg = model._level_cfg(model.nerf_mlp, 64, 1, 1.0, True)
            import ctypes as C
            assert tcfg.wgrad_mode == (hip.WGRAD_BF16X3 if hip.LEGACY_F16X2_TRAIN else hip.WGRAD_F16)      # one-half weight-gradient GEMM
            assert tcfg.precision == hip.PREC_F16X2 and int(hip.lib().refnerf_activations_format(C.byref(tcfg))) == (hip.ACT_F16X2 if hip.LEGACY_F16X2_TRAIN else hip.ACT_SQ)
    assert torch.equal(out["shipped"], out["f16x2"]) and not torch.equal(out["shipped"], out["f32"])
    assert float((out["shipped"] - out["f32"]).abs().max()) < 1e-4
    configs.clear_config()


def test_non_finite_training_loss_is_loud(hip):
    """weights that push hidden activations of the split-f16 chains beyond 65504 turn the level's outputs into NaN; the
    loss assembly (train_utils.compute_losses, Config.hip_check_finite) raises FloatingPointError one step later at the latest
    (its device flag is read without a synchronisation when the next step asks) and names the knobs that lift the limit; the
    exact-fp32 chains train through the same weights."""
    import os
    import torch
    from refnerf_pl_amd import configs, layout, models, synthetic, train_utils, utils
    P = synthetic.make_params(seed=0, bias_scale=0.05, sharpen=1.0)
    for name in ("spatial_net.2", "spatial_net.3"):
        sp = layout.SPEC_BY_NAME[name]
        P[sp.w_off:sp.w_off + sp.out_dim * sp.in_dim] *= 3000.0
    nxt = layout.SPEC_BY_NAME["spatial_net.4"]
    P[nxt.w_off:nxt.w_off + nxt.out_dim * nxt.in_dim] /= 9e6
    rays_np = synthetic.blender_rays(64, seed=2, center_frac=0.6)
    gt = synthetic.target_rgb(64, seed=5)
    for chains in ("f16x2", "f32"):
        configs.clear_config()
        configs.parse_config_files_and_bindings([os.path.join(os.path.dirname(__file__), "..", "configs", "refnerf_blender.gin")],
                                                ["Model.num_prop_samples = 64", "Model.num_nerf_samples = 64",
                                                 f"Config.hip_train_precision = '{chains}'", f"Config.hip_bwd_precision = '{chains}'"])
        cfg = configs.Config()
        model = models.construct_model(utils.dummy_rays(), cfg).to(DEV).train()
        model.nerf_mlp.load_flat_params(P)
        rays = utils.rays_from_dict(dict(rays_np), DEV)
        batch = utils.Batch(rays=rays, rgb=gt)
        train_utils.flush_finite_check(cfg)

        def step():
            rend, hist = model(rays, 1.0, False)
            return train_utils.compute_losses(model, batch, rays, rend, hist, cfg)[0]
        if chains == "f16x2":
            total = step()
            assert not bool(torch.isfinite(total))
            with pytest.raises(FloatingPointError, match="hip_train_precision"):
                step()                                   # the flag of the step before is read here
                train_utils.flush_finite_check(cfg)      # (not reached)
        else:
            assert bool(torch.isfinite(step())) and bool(torch.isfinite(step()))
            train_utils.flush_finite_check(cfg)
    configs.clear_config()
