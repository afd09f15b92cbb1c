"""ctypes binding of the CPU oracle (oracle/refnerf_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
Parity status: pinned against the reference's own outputs (tests/golden).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# oracle/oracle_f64.py executes this very file with _REAL_F64 set: the same binding over librefnerf_oracle_f64.so (the
# restatement compiled with the real type switched to double, refnerf_oracle_f64.h), float64 arrays in and out
_F64 = bool(globals().get("_REAL_F64", False))
_LIB_NAME = "librefnerf_oracle_f64.so" if _F64 else "librefnerf_oracle.so"
_LIB_PATH = os.path.join(_HERE, _LIB_NAME)
_REAL = C.c_double if _F64 else getattr(C, "c_float")
_NP = np.float64 if _F64 else getattr(np, "float32")
_SZ = 8 if _F64 else 4

WIDTH, DEPTH = 256, 8
SRGB_MODES = {"none": 0, "linear": 1, "norm_linear": 2, "srgb": 3, "norm_srgb": 4}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "refnerf_oracle.c")
    deps = [src, os.path.join(_HERE, "refnerf_oracle.h"), os.path.join(_HERE, "refnerf_oracle_f64.h"),
            os.path.join(os.path.dirname(_HERE), "include", "refnerf_detmath.h")]
    stale = (not os.path.exists(_LIB_PATH)
             or (os.path.exists(src) and os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(d) for d in deps if os.path.exists(d))))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", _LIB_NAME], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class LevelCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_samples", "n_in", "training", "compute_extras", "srgb_mapping",
        "srgb_mapping_normalization", "render_srgb_mode", "opaque_background",
        "ray_shape", "ide_mode", "raydist", "disable_integration")] + [(n, _REAL) for n in (
            "anneal", "resample_padding", "s_near", "s_far", "density_bias",
            "roughness_bias", "rgb_premultiplier", "rgb_bias", "rgb_padding", "bg_rgb")]


_FP = C.POINTER(_REAL)


class Rays(C.Structure):
    _fields_ = [(n, _FP) for n in ("origins", "directions", "viewdirs", "radii", "near", "far")]


_OUT_F32 = ("sdist",)
_OUT_FIELDS = [
    ("sdist", _FP), ("bin_idx", C.POINTER(C.c_int32)),
    ("density", _FP), ("rgb", _FP), ("normals", _FP), ("normals_pred", _FP),
    ("grad_pred", _FP), ("tint", _FP), ("diffuse", _FP), ("specular", _FP),
    ("roughness", _FP), ("weights", _FP),
    ("r_rgb", _FP), ("r_diffuse", _FP), ("r_specular", _FP), ("r_distance", _FP),
    ("r_acc", _FP), ("r_normals", _FP), ("r_normals_pred", _FP), ("r_tint", _FP),
    ("r_roughness", _FP), ("r_distance_mean", _FP), ("r_percentiles", C.POINTER(C.c_double)),
]


class LevelOut(C.Structure):
    _fields_ = _OUT_FIELDS


class SampleOut(C.Structure):
    _fields_ = [("density", _REAL), ("roughness", _REAL)] + [
        (n, _REAL * 3) for n in ("rgb", "normals", "normals_pred", "grad_pred", "tint", "diffuse", "specular")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.rn_level_forward.restype = C.c_int
        _lib.rn_s_to_t.restype = _REAL
        _lib.rn_s_to_t.argtypes = [_REAL] * 3
        _lib.rn_linear_to_srgb.restype = _REAL
        _lib.rn_linear_to_srgb.argtypes = [_REAL]
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=_NP)
    return a, a.ctypes.data_as(_FP)


def default_cfg(**kw) -> LevelCfg:
    c = LevelCfg()
    lib().rn_level_cfg_default(C.byref(c))
    for k, v in kw.items():
        if k == "render_srgb_mode" and isinstance(v, str):
            v = SRGB_MODES[v]
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def linspace_u(n):
    u = np.empty(n, _NP)
    lib().rn_linspace_u(C.c_int(n), u.ctypes.data_as(_FP))
    return u


def resample_logits(t, w, anneal=1.0, padding=0.01):
    t, tp = _f(t)
    w, wp = _f(w)
    out = np.empty(w.shape, _NP)
    M = w.shape[-1]
    for r in range(int(np.prod(w.shape[:-1], dtype=np.int64))):
        lib().rn_resample_logits(C.cast(C.addressof(tp.contents) + _SZ * r * (M + 1), _FP),
                                 C.cast(C.addressof(wp.contents) + _SZ * r * M, _FP), C.c_int(M),
                                 _REAL(anneal), _REAL(padding),
                                 C.cast(out.ctypes.data + _SZ * r * M, _FP))
    return out


def sample_intervals(t, w_logits, n, smin=0.0, smax=1.0):
    """t [R,M+1], w_logits [R,M] -> sdist [R,n+1], bin_idx [R,n]."""
    t, _ = _f(t)
    w_logits, _ = _f(w_logits)
    R, M = w_logits.shape
    sd = np.empty((R, n + 1), _NP)
    bi = np.empty((R, n), np.int32)
    for r in range(R):
        lib().rn_sample_intervals(t[r].ctypes.data_as(_FP), w_logits[r].ctypes.data_as(_FP),
                                  C.c_int(M), C.c_int(n), _REAL(smin), _REAL(smax),
                                  sd[r].ctypes.data_as(_FP), bi[r].ctypes.data_as(C.POINTER(C.c_int32)))
    return sd, bi


def cast_samples(origins, directions, radii, tdist, ray_shape=0):
    """-> lifted means [R,N,3], lifted vars [R,N,3], means xyz [R,N,3]."""
    o, _ = _f(origins)
    d, _ = _f(directions)
    rad = np.ascontiguousarray(radii, _NP).reshape(-1)
    td, _ = _f(tdist)
    R, N1 = td.shape
    lm = np.empty((R, N1 - 1, 3), _NP)
    lv = np.empty_like(lm)
    mx = np.empty_like(lm)
    for r in range(R):
        for i in range(N1 - 1):
            lib().rn_cast_sample(o[r].ctypes.data_as(_FP), d[r].ctypes.data_as(_FP), _REAL(rad[r]),
                                 _REAL(td[r, i]), _REAL(td[r, i + 1]), C.c_int(ray_shape),
                                 lm[r, i].ctypes.data_as(_FP), lv[r, i].ctypes.data_as(_FP),
                                 mx[r, i].ctypes.data_as(_FP))
    return lm, lv, mx


def ipe(lmean, lvar):
    lm, _ = _f(lmean)
    lv, _ = _f(lvar)
    flat_m, flat_v = lm.reshape(-1, 3), lv.reshape(-1, 3)
    out = np.empty((flat_m.shape[0], 96), _NP)
    for i in range(flat_m.shape[0]):
        lib().rn_ipe(flat_m[i].ctypes.data_as(_FP), flat_v[i].ctypes.data_as(_FP), out[i].ctypes.data_as(_FP))
    return out.reshape(lm.shape[:-1] + (96,))


def ide(xyz, kappa_inv, mode="stable"):
    """mode: 'stable' | 'ref32' | 'f64'."""
    kap = np.broadcast_to(np.asarray(kappa_inv, np.float64).reshape(-1), (np.asarray(xyz).reshape(-1, 3).shape[0],))
    if mode == "f64":
        x = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        out = np.empty((x.shape[0], 72), np.float64)
        for i in range(x.shape[0]):
            lib().rn_ide_f64(x[i].ctypes.data_as(C.POINTER(C.c_double)), C.c_double(kap[i]),
                             out[i].ctypes.data_as(C.POINTER(C.c_double)))
        return out
    fn = lib().rn_ide_stable_f32 if mode == "stable" else lib().rn_ide_ref_f32
    x = np.ascontiguousarray(xyz, _NP).reshape(-1, 3)
    out = np.empty((x.shape[0], 72), _NP)
    for i in range(x.shape[0]):
        fn(x[i].ctypes.data_as(_FP), _REAL(kap[i]), out[i].ctypes.data_as(_FP))
    return out


def mlp_samples(params, cfg, lmean, lvar, viewdirs):
    """Per-sample MLP.__call__; lmean/lvar [S,3], viewdirs [S,3] -> dict of arrays."""
    p, pp = _f(params)
    lm, _ = _f(lmean)
    lv, _ = _f(lvar)
    v, _ = _f(viewdirs)
    S = lm.shape[0]
    names = ("rgb", "normals", "normals_pred", "grad_pred", "tint", "diffuse", "specular")
    res = {n: np.empty((S, 3), _NP) for n in names}
    res["density"] = np.empty(S, _NP)
    res["roughness"] = np.empty(S, _NP)
    so = SampleOut()
    for i in range(S):
        lib().rn_mlp_sample(pp, C.byref(cfg), lm[i].ctypes.data_as(_FP), lv[i].ctypes.data_as(_FP),
                            v[i].ctypes.data_as(_FP), C.byref(so))
        res["density"][i] = so.density
        res["roughness"][i] = so.roughness
        for n in names:
            res[n][i] = np.array(getattr(so, n)[:], _NP)
    return res


def alpha_weights(density, tdist, dirs, opaque_background=False):
    dn, _ = _f(density)
    td, _ = _f(tdist)
    d, _ = _f(dirs)
    R, N = dn.shape
    out = np.empty((R, N), _NP)
    for r in range(R):
        lib().rn_alpha_weights(dn[r].ctypes.data_as(_FP), td[r].ctypes.data_as(_FP), d[r].ctypes.data_as(_FP),
                               C.c_int(N), C.c_int(int(opaque_background)), out[r].ctypes.data_as(_FP))
    return out


def render_rays(density, tdist, dirs, far, rgb=None, diffuse=None, specular=None, normals=None, normals_pred=None,
                roughness=None, tint=None, **cfg_kw):
    """rn_render_rays: compute_alpha_weights + volumetric_rendering (render.py:132-254) on caller-supplied
    per-sample values -> dict(weights, r_rgb, r_diffuse, r_specular, r_distance, r_acc, extras, r_percentiles)."""
    lib().rn_render_rays.restype = C.c_int
    dn, dnp = _f(density)
    R, N = dn.shape
    td, tdp = _f(np.asarray(tdist).reshape(R, N + 1))
    d, dp = _f(np.asarray(dirs).reshape(R, 3))
    fr, frp = _f(np.asarray(far).reshape(R))
    opt, keep = [], []
    for a, last in ((rgb, 3), (diffuse, 3), (specular, 3), (normals, 3), (normals_pred, 3), (roughness, 1), (tint, 3)):
        if a is None:
            opt.append(None)
        else:
            arr, ap = _f(np.asarray(a).reshape(R, N, last))
            keep.append(arr)
            opt.append(ap)
    cfg = default_cfg(n_samples=N, training=int(normals is not None), **cfg_kw)
    out = LevelOut()
    res = {"weights": np.zeros((R, N), _NP), "r_rgb": np.zeros((R, 3), _NP),
           "r_diffuse": np.zeros((R, 3), _NP), "r_specular": np.zeros((R, 3), _NP),
           "r_distance": np.zeros(R, _NP), "r_acc": np.zeros(R, _NP),
           "r_normals": np.zeros((R, 3), _NP), "r_normals_pred": np.zeros((R, 3), _NP),
           "r_tint": np.zeros((R, 3), _NP), "r_roughness": np.zeros(R, _NP),
           "r_distance_mean": np.zeros(R, _NP)}
    for k, a in res.items():
        setattr(out, k, a.ctypes.data_as(_FP))
    res["r_percentiles"] = np.zeros((R, 3), np.float64)
    out.r_percentiles = res["r_percentiles"].ctypes.data_as(C.POINTER(C.c_double))
    rc = lib().rn_render_rays(C.byref(cfg), C.c_int(R), dnp, tdp, dp, frp, *opt, C.byref(out))
    if rc != 0:
        raise ValueError(f"rn_render_rays failed with code {rc}")
    return res


def _rays_struct(rays: dict):
    keep = {}
    rs = Rays()
    for name in ("origins", "directions", "viewdirs", "radii", "near", "far"):
        arr = np.ascontiguousarray(np.asarray(rays[name], _NP))
        keep[name] = arr
        setattr(rs, name, arr.ctypes.data_as(_FP))
    return rs, keep


def level_forward(params, cfg: LevelCfg, rays: dict, sdist_in, weights_in, n_threads=0, history=True):
    """One level of Model.__call__ for R rays -> dict of numpy outputs."""
    p, pp = _f(params)
    rs, keep = _rays_struct(rays)
    R = keep["origins"].shape[0]
    N, M = cfg.n_samples, cfg.n_in
    sd_in, sdp = _f(np.asarray(sdist_in).reshape(R, M + 1))
    w_in, wp = _f(np.asarray(weights_in).reshape(R, M))
    out = LevelOut()
    res = {}
    shapes = {
        "sdist": (R, N + 1), "density": (R, N), "rgb": (R, N, 3), "normals": (R, N, 3),
        "normals_pred": (R, N, 3), "grad_pred": (R, N, 3), "tint": (R, N, 3), "diffuse": (R, N, 3),
        "specular": (R, N, 3), "roughness": (R, N), "weights": (R, N),
        "r_rgb": (R, 3), "r_diffuse": (R, 3), "r_specular": (R, 3), "r_distance": (R,), "r_acc": (R,),
        "r_normals": (R, 3), "r_normals_pred": (R, 3), "r_tint": (R, 3), "r_roughness": (R,),
        "r_distance_mean": (R,),
    }
    per_sample = {"density", "rgb", "normals", "normals_pred", "grad_pred", "tint", "diffuse", "specular", "roughness"}
    for name, shp in shapes.items():
        if not history and name in per_sample:
            continue
        if name in ("normals", "r_normals") and not cfg.training:
            continue
        res[name] = np.zeros(shp, _NP)
        setattr(out, name, res[name].ctypes.data_as(_FP))
    res["bin_idx"] = np.zeros((R, N), np.int32)
    out.bin_idx = res["bin_idx"].ctypes.data_as(C.POINTER(C.c_int32))
    if cfg.compute_extras:
        res["r_percentiles"] = np.zeros((R, 3), np.float64)
        out.r_percentiles = res["r_percentiles"].ctypes.data_as(C.POINTER(C.c_double))
    rc = lib().rn_level_forward(pp, C.byref(cfg), C.byref(rs), C.c_int(R), sdp, wp, C.byref(out), C.c_int(n_threads))
    if rc != 0:
        raise ValueError(f"rn_level_forward failed with code {rc}")
    return res


def model_forward(params, rays: dict, num_levels=2, num_prop_samples=128, num_nerf_samples=128,
                  n_threads=0, history=True, **cfg_kw):
    """Model.__call__ (models.py:129-321): the level loop over level_forward."""
    R = np.asarray(rays["origins"]).shape[0]
    s_near = cfg_kw.get("s_near", 0.0)
    s_far = cfg_kw.get("s_far", 1.0)
    sdist = np.tile(np.array([[s_near, s_far]], _NP), (R, 1))
    weights = np.ones((R, 1), _NP)
    outs = []
    for lvl in range(num_levels):
        n = num_prop_samples if lvl < num_levels - 1 else num_nerf_samples
        cfg = default_cfg(n_samples=n, n_in=weights.shape[1], **cfg_kw)
        res = level_forward(params, cfg, rays, sdist, weights, n_threads=n_threads, history=history)
        outs.append(res)
        sdist, weights = res["sdist"], res["weights"]
    return outs


class LossCfg(C.Structure):
    _fields_ = [("data_mult", _REAL), ("orientation_mult", _REAL), ("normal_mult", _REAL)]


def model_train(params, rays: dict, gt_rgb, num_levels=2, num_prop_samples=128, num_nerf_samples=128,
                data_mults=(0.1, 1.0), orientation_mults=(0.01, 0.1), normal_mults=(3e-5, 3e-4),
                n_threads=0, **cfg_kw):
    """Forward + data/orientation/predicted-normal losses + backward of Model.__call__
    in training mode.  *_mults = (coarse, fine) multipliers (blender_refnerf.gin:9-16).
    Returns (losses dict, grads blob float32, per-level dicts with sdist/weights/r_rgb)."""
    lib().rn_level_train.restype = C.c_int
    p, pp = _f(params)
    rs, keep = _rays_struct(rays)
    R = keep["origins"].shape[0]
    gt, gtp = _f(np.asarray(gt_rgb)[..., :3].reshape(R, 3))
    lm, lmp = _f(np.asarray(rays["lossmult"]).reshape(R))
    grads = np.zeros(p.shape[0], _NP)
    sdist = np.tile(np.array([[cfg_kw.get("s_near", 0.0), cfg_kw.get("s_far", 1.0)]], _NP), (R, 1))
    weights = np.ones((R, 1), _NP)
    losses = {"data": 0.0, "orientation": 0.0, "normal": 0.0}
    levels = []
    for lvl in range(num_levels):
        fine = lvl == num_levels - 1
        n = num_nerf_samples if fine else num_prop_samples
        cfg = default_cfg(n_samples=n, n_in=weights.shape[1], training=1, **cfg_kw)
        lc = LossCfg(data_mults[1] if fine else data_mults[0], orientation_mults[1] if fine else orientation_mults[0],
                     normal_mults[1] if fine else normal_mults[0])
        out = LevelOut()
        res = {"sdist": np.zeros((R, n + 1), _NP), "weights": np.zeros((R, n), _NP),
               "r_rgb": np.zeros((R, 3), _NP)}
        for k, a in res.items():
            setattr(out, k, a.ctypes.data_as(_FP))
        loss3 = (C.c_double * 3)()
        sd_in, sdp = _f(sdist)
        w_in, wp = _f(weights)
        rc = lib().rn_level_train(pp, C.byref(cfg), C.byref(rs), C.c_int(R), sdp, wp, gtp, lmp, C.byref(lc), C.byref(out),
                                  grads.ctypes.data_as(_FP), loss3, C.c_int(n_threads))
        if rc != 0:
            raise ValueError(f"rn_level_train failed with code {rc}")
        losses["data"] += loss3[0]
        losses["orientation"] += loss3[1]
        losses["normal"] += loss3[2]
        levels.append(res)
        sdist, weights = res["sdist"], res["weights"]
    losses["total"] = losses["data"] + losses["orientation"] + losses["normal"]
    return losses, grads, levels


_SEED_FIELDS = ("g_r_rgb", "g_r_diffuse", "g_r_specular", "g_r_acc", "g_r_distance", "g_r_normals",
                "g_r_normals_pred", "g_r_tint", "g_r_roughness", "g_weights", "g_density", "g_roughness",
                "g_rgb", "g_normals_pred", "g_tint", "g_diffuse", "g_specular")


class LevelSeeds(C.Structure):
    _fields_ = [(n, _FP) for n in _SEED_FIELDS]


def level_backward(params, cfg: LevelCfg, rays: dict, sdist_in, weights_in, seeds: dict, grads=None, n_threads=0):
    """rn_level_backward: upstream gradients on any of the level's differentiable outputs
    (keys of _SEED_FIELDS without the "g_" prefix: r_* per ray, the others per sample) ->
    accumulated into `grads` (canonical blob, float32)."""
    lib().rn_level_backward.restype = C.c_int
    p, pp = _f(params)
    rs, keep = _rays_struct(rays)
    R = keep["origins"].shape[0]
    N, M = cfg.n_samples, cfg.n_in
    sd_in, sdp = _f(np.asarray(sdist_in).reshape(R, M + 1))
    w_in, wp = _f(np.asarray(weights_in).reshape(R, M))
    st = LevelSeeds()
    hold = []
    for k, v in seeds.items():
        if v is None:
            continue
        if "g_" + k not in _SEED_FIELDS:
            raise KeyError(k)
        a, ap = _f(v)
        hold.append(a)
        setattr(st, "g_" + k, ap)
    if grads is None:
        grads = np.zeros(p.shape[0], _NP)
    rc = lib().rn_level_backward(pp, C.byref(cfg), C.byref(rs), C.c_int(R), sdp, wp, C.byref(st),
                                 grads.ctypes.data_as(_FP), C.c_int(n_threads))
    if rc != 0:
        raise ValueError(f"rn_level_backward failed with code {rc}")
    return grads
