"""Shared helpers for the parity tests."""
import os

import numpy as np

import refnerf_pl_amd  # noqa: F401
from refnerf_pl_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def trained_blob():
    """The "trained-like" weights: the blob the REFERENCE reached after 400 of its own Adam steps on an analytic
    shiny sphere (tests/golden/make_golden.py::golden_trained), stored as float16; the float16 -> float32 upcast is
    the fixture's weight set (the reference outputs were computed from exactly these values)."""
    return np.load(os.path.join(GOLDEN, "trained_blob.npz"))["blob_f16"].astype(np.float32)


def trained_long_blob():
    """The harsher trained-like weights: the blob the REFERENCE reached after 2500 of its own Adam steps at lr 1e-3
    (tests/golden/make_golden.py::golden_trained_long), stored in float32 -- off every 16-bit grid."""
    return np.load(os.path.join(GOLDEN, "trained_long_blob.npz"))["blob_f32"]


def trained_llff_blob():
    """Trained-like weights of the forward-facing configurations: 1200 of the REFERENCE's Adam steps on an analytic NDC scene
    (linear colour, norm_linear render map; tests/golden/make_golden.py::golden_trained_llff), float32."""
    return np.load(os.path.join(GOLDEN, "trained_llff_blob.npz"))["blob_f32"]


def params_from_golden(g):
    pk = g["param_kw"]
    if int(pk[0]) == -3:
        return trained_llff_blob()
    if int(pk[0]) == -2:
        return trained_long_blob()
    if int(pk[0]) < 0:
        return trained_blob()
    return synthetic.make_params(int(pk[0]), float(pk[1]), float(pk[2]), float(pk[3]))


def rays_from_golden(g):
    return {k[5:]: g[k] for k in g.files if k.startswith("rays_")}


def cfg_from_bindings(bindings):
    """Translate the gin bindings stored in a model fixture into
    (level-cfg kwargs, Model kwargs)."""
    kw, lv = {}, {}
    for s in bindings:
        s = str(s)
        if "NerfMLP.srgb_mapping = False" in s:
            kw["srgb_mapping"] = 0
        if "srgb_mapping_type" in s:
            kw["render_srgb_mode"] = s.split("'")[1]
        for key in ("num_levels", "num_nerf_samples", "num_prop_samples"):
            if "Model." + key in s:
                lv[key] = int(s.split("=")[1])
    return kw, lv


MODEL_CASES = ["model_blender_eval", "model_blender_sharp_eval", "model_c1_eval",
               "model_llff_linear_eval", "model_blender_sharp_train", "model_llff_linear_train",
               "model_shiny_eval", "model_shiny_train", "model_trained_eval", "model_trained_train"]
EVAL_CASES = [n for n in MODEL_CASES if n.endswith("eval")]
TRAIN_CASES = [n for n in MODEL_CASES if n.endswith("train")]

HIST_KEYS = ("sdist", "weights", "density", "rgb", "normals_pred", "roughness", "diffuse", "specular", "tint")
REND_KEYS = ("rgb", "diffuse", "specular", "distance", "acc", "normals_pred", "tint", "roughness", "distance_mean")


# ---- NerfMLP variants served by embedding (refnerf_pl_amd.layout.variant_layout) ------------------------------------
VARIANT_CASES = ["model_variant_eval", "model_variant_train", "model_variant_nonormals_train"]
VARIANT_KW = dict(net_width_viewdirs=128, use_n_dot_v=False, use_specular_tint=False, enable_pred_roughness=False)
VARIANT_HIST_KEYS = ("sdist", "weights", "density", "rgb", "normals_pred", "diffuse", "specular")
VARIANT_REND_KEYS = ("rgb", "diffuse", "specular", "distance", "acc", "normals_pred", "distance_mean")


def variant_params(g):
    """(canonical blob with the variant embedded, the variant's own flat blob, index): the fixture's weights are the
    embedded elements of the synthetic canonical blob its param_kw names."""
    from refnerf_pl_amd import layout
    full = params_from_golden(g)
    _, idx = layout.variant_layout(**VARIANT_KW)
    canon = np.zeros_like(full)
    canon[idx] = full[idx]
    return canon, full[idx].copy(), idx


POSENC_CASES = ["model_posenc_eval", "model_posenc_train"]


def posenc_params(g):
    """as variant_params, for `use_directional_enc = False` (deg_view 5, everything else Ref-NeRF)"""
    from refnerf_pl_amd import layout
    full = params_from_golden(g)
    _, idx = layout.variant_layout(use_directional_enc=False, deg_view=5)
    canon = np.zeros_like(full)
    canon[idx] = full[idx]
    return canon, full[idx].copy(), idx


# ---- Model.raydist_fn / Model.disable_integration (coord.py:63-99, models.py:228-231) ------------------------------
RAYDIST_CASES = ["model_raydist_reciprocal_eval", "model_raydist_log_eval", "model_raydist_piecewise_eval",
                 "model_nointegration_eval", "model_raydist_nointegration_train"]
RAYDIST_CODE = {"": 0, "piecewise": 1, "reciprocal": 2, "log": 3, "exp": 4, "sqrt": 5, "square": 6}


def raydist_kw(g):
    """level-cfg kwargs of a raydist fixture"""
    return dict(raydist=RAYDIST_CODE[str(g["raydist_fn"])], disable_integration=int(g["disable_integration"]))
