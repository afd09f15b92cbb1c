"""ctypes binding of librefnerf_hip.so (include/refnerf_hip.h).

PyTorch is used only for device memory and streams: every call takes raw
device pointers (``tensor.data_ptr()``) and the current HIP stream.  There is
no CPU fallback -- if the library or a gfx950 device is missing the call
raises.
"""
import ctypes as C
import os
import subprocess

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "librefnerf_hip.so")
if os.environ.get("REFNERF_LIB"):      # A/B builds of the library (scripts/build_*_variant.sh -> ab/*.so): a debug knob, never a fallback
    LIB_PATH = os.path.abspath(os.environ["REFNERF_LIB"])

PREC_F32, PREC_BF16, PREC_F16, PREC_F16X2 = 0, 1, 2, 3
IMAGE_F16X2_TRAIN = 4   # REFNERF_IMAGE_F16X2_TRAIN: the weight image of the REFNERF_PREC_F16X2 training kernels (built-in basis)
ACT_F32, ACT_BF16, ACT_F16X2, ACT_SQ = 0, 1, 2, 3   # REFNERF_ACT_*
ABI_VERSION = 11  # REFNERF_ABI_VERSION
# debug knob, read once by the library as well: the round-4 training kernels of the f16x2 mode (REFNERF_ACT_F16X2, f32 image)
LEGACY_F16X2_TRAIN = os.environ.get("REFNERF_LEGACY_F16X2_TRAIN", "0") not in ("", "0")
WGRAD_F32, WGRAD_BF16X3, WGRAD_F16 = 0, 1, 2
DIRENC_IDE, DIRENC_POSENC = 0, 1   # REFNERF_DIRENC_*
RAYDIST = {None: 0, "piecewise": 1, "reciprocal": 2, "log": 3, "exp": 4, "sqrt": 5, "square": 6}   # REFNERF_RAYDIST_*
SRGB_MODES = {"none": 0, "linear": 1, "norm_linear": 2, "srgb": 3, "norm_srgb": 4}

_FP = C.c_void_p
NUM_PARAMS = 1110158   # REFNERF_NUM_PARAMS
NUM_PARAMS_EXT = NUM_PARAMS + 2 * 6 * 256 * 96   # REFNERF_NUM_PARAMS_EXT (general IPE basis: canonical blob + group tail)
IPE_MAX_GROUPS = 7


class LevelCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_samples", "n_in", "training", "compute_extras", "srgb_mapping",
        "srgb_mapping_normalization", "render_srgb_mode", "opaque_background",
        "ray_shape", "precision", "wgrad_mode", "dir_enc", "raydist", "disable_integration", "ipe_groups")] + [(n, C.c_float) for n in (
            "anneal", "resample_padding", "s_near", "s_far", "density_bias",
            "roughness_bias", "rgb_premultiplier", "rgb_bias", "rgb_padding", "bg_rgb")]


class RaysStruct(C.Structure):
    _fields_ = [(n, _FP) for n in ("d_origins", "d_directions", "d_viewdirs", "d_radii", "d_near", "d_far")]


OUT_FIELDS = ("d_sdist", "d_bin_idx", "d_density", "d_rgb", "d_normals", "d_normals_pred", "d_grad_pred",
              "d_tint", "d_diffuse", "d_specular", "d_roughness", "d_weights", "d_r_rgb", "d_r_diffuse",
              "d_r_specular", "d_r_distance", "d_r_acc", "d_r_normals", "d_r_normals_pred", "d_r_tint",
              "d_r_roughness", "d_r_distance_mean", "d_r_percentiles")


class LevelOut(C.Structure):
    _fields_ = [(n, _FP) for n in OUT_FIELDS]


class LevelSaved(C.Structure):
    _fields_ = [(n, _FP) for n in ("d_sdist", "d_density", "d_rgb", "d_weights", "d_activations")] + [
        ("activations_format", C.c_int32)]


class LevelGrads(C.Structure):
    _fields_ = [(n, _FP) for n in ("d_g_r_rgb", "d_g_weights", "d_g_normals_pred", "d_g_r_acc", "d_g_r_distance",
                                   "d_g_density", "d_g_rgb", "d_g_diffuse", "d_g_specular", "d_g_tint",
                                   "d_g_roughness")]


class HipLibraryError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the library in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC] + (["-B"] if force else []) + ["librefnerf_hip.so"]
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    """Load the library; raises HipLibraryError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the Ref-NeRF hot path has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.refnerf_last_error.restype = C.c_char_p
        L.refnerf_packed_weights_bytes.restype = C.c_size_t
        L.refnerf_packed_weights_bytes.argtypes = [C.c_int]
        L.refnerf_pack_weights.argtypes = [_FP, _FP, C.c_int, _FP]
        L.refnerf_packed_weights_bytes_basis.restype = C.c_size_t
        L.refnerf_packed_weights_bytes_basis.argtypes = [C.c_int, C.c_int]
        L.refnerf_pack_weights_basis.argtypes = [_FP, _FP, C.c_int, _FP, C.c_int, _FP]
        L.refnerf_level_forward.argtypes = [_FP, C.POINTER(LevelCfg), C.POINTER(RaysStruct), C.c_int32,
                                            _FP, _FP, C.POINTER(LevelOut), _FP]
        L.refnerf_activations_format.argtypes = [C.POINTER(LevelCfg)]
        L.refnerf_level_image.argtypes = [C.POINTER(LevelCfg)]
        L.refnerf_level_image.restype = C.c_int
        L.refnerf_activation_workspace_bytes.restype = C.c_size_t
        L.refnerf_activation_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
        L.refnerf_level_forward_train.argtypes = [_FP, C.POINTER(LevelCfg), C.POINTER(RaysStruct), C.c_int32,
                                                  _FP, _FP, C.POINTER(LevelOut), _FP, C.c_size_t, _FP]
        L.refnerf_backward_workspace_bytes.restype = C.c_size_t
        L.refnerf_backward_workspace_bytes_basis.restype = C.c_size_t
        L.refnerf_backward_workspace_bytes_basis.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.refnerf_activation_workspace_bytes_basis.restype = C.c_size_t
        L.refnerf_activation_workspace_bytes_basis.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.refnerf_backward_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
        L.refnerf_level_backward.argtypes = [_FP, C.POINTER(LevelCfg), C.POINTER(RaysStruct), C.c_int32,
                                             C.POINTER(LevelSaved), C.POINTER(LevelGrads), _FP, _FP, C.c_size_t, _FP]
        L.refnerf_pixels_to_rays.argtypes = [_FP, _FP, _FP, C.c_int32, _FP, C.c_int32, _FP, C.c_int32,
                                             _FP, _FP, _FP, _FP, _FP, _FP]
        L.refnerf_mlp_forward.argtypes = [_FP, C.POINTER(LevelCfg), _FP, _FP, C.c_int32, _FP, C.c_int32, C.c_int32,
                                          C.POINTER(LevelOut), _FP]
        L.refnerf_sample_intervals.argtypes = [_FP, _FP, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                               _FP, _FP, _FP]
        L.refnerf_integrated_pos_enc.argtypes = [_FP, _FP, C.c_int32, _FP, _FP]
        L.refnerf_integrated_dir_enc.argtypes = [_FP, _FP, C.c_int32, _FP, _FP]
        L.refnerf_render_rays.argtypes = [C.POINTER(LevelCfg), C.c_int32] + [_FP] * 11 + [C.POINTER(LevelOut), _FP]
        L.refnerf_losses_forward.argtypes = [C.c_int32, C.c_int32] + [_FP] * 9 + [_FP]
        L.refnerf_losses_backward.argtypes = [C.c_int32, C.c_int32] + [_FP] * 5 + [C.c_int32] + [_FP] * 3 + [C.c_float] * 3 + [_FP] * 4 + [_FP]
        L.refnerf_get_timing.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.refnerf_get_timing_family.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        if L.refnerf_abi_version() != ABI_VERSION:
            raise HipLibraryError("librefnerf_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int):
    if rc == 0:
        return
    msg = lib().refnerf_last_error().decode()
    if rc == -1:
        raise ValueError(msg)      # same exception type the reference raises for bad arguments
    raise HipLibraryError(f"librefnerf_hip error {rc}: {msg}")


def require_device():
    """Fail loudly unless the current device is an MI355X-class (gfx950) GPU."""
    if not torch.cuda.is_available():
        raise HipLibraryError("no HIP device visible: the Ref-NeRF hot path needs a gfx950 GPU (no CPU fallback)")
    check(lib().refnerf_device_ok())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return C.c_void_p(t.data_ptr())


def default_cfg(**kw) -> LevelCfg:
    c = LevelCfg()
    lib().refnerf_level_cfg_default(C.byref(c))
    for k, v in kw.items():
        if k == "render_srgb_mode" and isinstance(v, str):
            v = SRGB_MODES[v]
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def level_image(precision: int, training: bool = False, ipe_groups: int = 0) -> int:
    """Which weight image refnerf_level_forward / _forward_train / _backward expect as `d_packed` for a level configuration:
    inference levels the image of their precision mode (a general IPE basis: the f32 image), training levels the f32 image --
    except REFNERF_PREC_F16X2 on the built-in basis, whose training kernels stream their own image (REFNERF_IMAGE_F16X2_TRAIN)."""
    # one rule, inside the library (ABI v11: refnerf_level_image; it reads the REFNERF_LEGACY_F16X2_TRAIN switch itself)
    cfg = LevelCfg()
    cfg.precision, cfg.training, cfg.ipe_groups = int(precision), int(bool(training)), int(ipe_groups)
    return int(lib().refnerf_level_image(C.byref(cfg)))


def packed_weights_bytes(precision=PREC_F32) -> int:
    return int(lib().refnerf_packed_weights_bytes(precision))


def pack_weights(params: torch.Tensor, packed: torch.Tensor = None, precision=PREC_F32, basis: torch.Tensor = None) -> torch.Tensor:
    """canonical fp32 blob (device) -> MFMA operand image (device).  `basis` ([3 G, 3] fp32 device tensor, G = 2..7: a general
    IPE basis, refnerf_pack_weights_basis): `params` is then the extended blob (NUM_PARAMS_EXT)."""
    require_device()
    groups = 0 if basis is None else basis.shape[0] // 3
    assert params.dtype == torch.float32 and params.numel() == (NUM_PARAMS_EXT if groups > 1 else NUM_PARAMS)
    nbytes = int(lib().refnerf_packed_weights_bytes_basis(precision, groups))
    if nbytes == 0:
        if groups > 1:
            raise ValueError("a general IPE basis (NerfMLP.basis_shape / basis_subdivisions other than 'octahedron' / 1) has at most 21 "
                             "directions and runs on the f32 operand image ('f32' and 'f16x2' modes); the plain bf16 / f16 images have "
                             "no direction groups")
        raise HipLibraryError("precision mode not built")
    if packed is None or packed.numel() * packed.element_size() != nbytes or packed.device != params.device:
        packed = torch.empty(nbytes // 4, dtype=torch.float32, device=params.device)
    if groups > 1:
        assert basis.dtype == torch.float32 and basis.is_contiguous() and basis.device == params.device and basis.shape == (3 * groups, 3)
        check(lib().refnerf_pack_weights_basis(ptr(params), ptr(basis), groups, ptr(packed), precision, stream_ptr()))
    else:
        check(lib().refnerf_pack_weights(ptr(params), ptr(packed), precision, stream_ptr()))
    return packed


def level_forward(packed, cfg: LevelCfg, rays: dict, sdist_in, weights_in, history=True, save_activations=False):
    """One fused level.  rays: dict of device tensors (origins, directions,
    viewdirs [R,3]; radii, near, far [R] or [R,1]).  Returns dict of tensors.
    save_activations (training forward): also keeps the layer inputs for
    level_backward in res["activations"] (a byte tensor the backward consumes)."""
    require_device()
    dev = sdist_in.device
    R = rays["origins"].shape[0]
    N = cfg.n_samples
    rs = RaysStruct()
    keep = []
    for name in ("origins", "directions", "viewdirs", "radii", "near", "far"):
        t = rays[name].to(torch.float32).contiguous()
        keep.append(t)
        setattr(rs, "d_" + name, t.data_ptr())
    f32 = dict(dtype=torch.float32, device=dev)
    res = {
        "sdist": torch.empty((R, N + 1), **f32), "bin_idx": torch.empty((R, N), dtype=torch.int32, device=dev),
        "weights": torch.empty((R, N), **f32),
        "r_rgb": torch.empty((R, 3), **f32), "r_diffuse": torch.empty((R, 3), **f32),
        "r_specular": torch.empty((R, 3), **f32), "r_distance": torch.empty((R,), **f32),
        "r_acc": torch.empty((R,), **f32),
    }
    # history: True = every per-sample output (the reference's ray_history), False = none, or the names wanted
    want = ("rgb", "normals_pred", "grad_pred", "tint", "diffuse", "specular", "density", "roughness", "normals") \
        if history is True else (tuple(history) if history else ())
    for k in ("rgb", "normals_pred", "grad_pred", "tint", "diffuse", "specular"):
        if k in want:
            res[k] = torch.empty((R, N, 3), **f32)
    for k in ("density", "roughness"):
        if k in want:
            res[k] = torch.empty((R, N), **f32)
    if cfg.training and "normals" in want:
        res["normals"] = torch.empty((R, N, 3), **f32)
    if cfg.compute_extras:
        res["r_normals_pred"] = torch.empty((R, 3), **f32)
        res["r_tint"] = torch.empty((R, 3), **f32)
        res["r_roughness"] = torch.empty((R,), **f32)
        res["r_distance_mean"] = torch.empty((R,), **f32)
        res["r_percentiles"] = torch.empty((R, 3), dtype=torch.float64, device=dev)
        if cfg.training:
            res["r_normals"] = torch.empty((R, 3), **f32)
    out = LevelOut()
    for k, t in res.items():
        setattr(out, "d_" + k, t.data_ptr())
    sd = sdist_in.to(torch.float32).contiguous()
    w = weights_in.to(torch.float32).contiguous()
    if save_activations:
        act = torch.empty(lib().refnerf_activation_workspace_bytes_basis(R, N, cfg.ipe_groups), dtype=torch.uint8, device=dev)
        check(lib().refnerf_level_forward_train(ptr(packed), C.byref(cfg), C.byref(rs), R, ptr(sd), ptr(w), C.byref(out),
                                                ptr(act), act.numel(), stream_ptr()))
        res["activations"] = act
        # REFNERF_ACT_*: what this forward wrote (f32 rows | bf16 pair-rows | split-f16 hi / lo pair units), as the library says
        res["activations_format"] = int(lib().refnerf_activations_format(C.byref(cfg)))
    else:
        check(lib().refnerf_level_forward(ptr(packed), C.byref(cfg), C.byref(rs), R, ptr(sd), ptr(w), C.byref(out), stream_ptr()))
    return res


def pixels_to_rays(pix_x, pix_y, pixtocams, camtoworlds, pixtocam_ndc=None):
    """Device tensors in, device tensors out: (origins, directions, viewdirs [n,3], radii [n,1], imageplane [n,2])."""
    require_device()
    dev = pix_x.device
    n = pix_x.numel()
    px = pix_x.reshape(-1).to(torch.int32).contiguous()
    py = pix_y.reshape(-1).to(torch.int32).contiguous()
    p2c = pixtocams.to(torch.float32).contiguous()
    c2w = camtoworlds.to(torch.float32)[..., :3, :4].contiguous()
    ndc = None if pixtocam_ndc is None else pixtocam_ndc.to(torch.float32).contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    o, d, v = (torch.empty((n, 3), **f32) for _ in range(3))
    r = torch.empty((n, 1), **f32)
    ip = torch.empty((n, 2), **f32)
    check(lib().refnerf_pixels_to_rays(ptr(px), ptr(py), ptr(p2c), int(p2c.dim() > 2), ptr(c2w), int(c2w.dim() > 2),
                                       ptr(ndc) if ndc is not None else None, n, ptr(o), ptr(d), ptr(v), ptr(r), ptr(ip),
                                       stream_ptr()))
    return o, d, v, r, ip


def mlp_forward(packed, cfg: LevelCfg, means, covs, viewdirs):
    """MLP.__call__ on caller-supplied Gaussians: means [R,N,3], covs [R,N,3,3] or [R,N,3], viewdirs [R,3]
    (device tensors) -> dict of per-sample tensors (the reference's `ray_results`)."""
    require_device()
    R, N = means.shape[0], means.shape[1]
    f32 = dict(dtype=torch.float32, device=means.device)
    m = means.to(torch.float32).contiguous()
    c = covs.to(torch.float32).contiguous()
    v = viewdirs.to(torch.float32).contiguous()
    full = 1 if c.dim() == 4 else 0
    res = {k: torch.empty((R, N, 3), **f32) for k in ("rgb", "normals_pred", "grad_pred", "tint", "diffuse", "specular")}
    res["density"] = torch.empty((R, N), **f32)
    res["roughness"] = torch.empty((R, N), **f32)
    if cfg.training:
        res["normals"] = torch.empty((R, N, 3), **f32)
    out = LevelOut()
    for k, t in res.items():
        setattr(out, "d_" + k, t.data_ptr())
    check(lib().refnerf_mlp_forward(ptr(packed), C.byref(cfg), ptr(m), ptr(c), full, ptr(v), R, N, C.byref(out), stream_ptr()))
    return res


_workspace = {}


def backward_workspace(R: int, n_samples: int, device, ipe_groups: int = 0) -> torch.Tensor:
    """Cached byte workspace for refnerf_level_backward (grown on demand, one per device)."""
    need = lib().refnerf_backward_workspace_bytes_basis(R, n_samples, ipe_groups)
    ws = _workspace.get(device)
    if ws is None or ws.numel() < need:
        _workspace[device] = None
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _workspace[device] = ws
    return ws


def level_backward(packed, cfg: LevelCfg, rays: dict, saved: dict, g_r_rgb, g_weights, g_normals_pred,
                   param_grads: torch.Tensor, g_r_acc=None, g_r_distance=None, sample_seeds=None):
    """Backward of one level: accumulates dL/d(params) into `param_grads`
    (canonical blob).  saved: dict with sdist, density, rgb, weights and
    activations of the training forward (save_activations=True); g_*: upstream
    gradients (g_weights / g_normals_pred may be None); sample_seeds: optional dict of
    per-sample seeds on ray_history {density, rgb, diffuse, specular, tint, roughness}."""
    require_device()
    R = rays["origins"].shape[0]
    N = cfg.n_samples
    rs = RaysStruct()
    keep = []
    for name in ("origins", "directions", "viewdirs", "radii", "near", "far"):
        t = rays[name].to(torch.float32).contiguous()
        keep.append(t)
        setattr(rs, "d_" + name, t.data_ptr())
    sv = LevelSaved()
    for name in ("sdist", "density", "rgb", "weights"):
        t = saved[name].to(torch.float32).contiguous()
        keep.append(t)
        setattr(sv, "d_" + name, t.data_ptr())
    if saved.get("activations") is None:
        raise ValueError("level_backward needs the activations saved by level_forward(..., save_activations=True)")
    sv.d_activations = saved["activations"].data_ptr()
    sv.activations_format = int(saved.get("activations_format", 0))      # REFNERF_ACT_*: the forward's precision
    gr = LevelGrads()
    for name, t in (("d_g_r_rgb", g_r_rgb), ("d_g_weights", g_weights), ("d_g_normals_pred", g_normals_pred),
                    ("d_g_r_acc", g_r_acc), ("d_g_r_distance", g_r_distance)) + tuple(
                        ("d_g_" + k, v) for k, v in (sample_seeds or {}).items()):
        if t is not None:
            t = t.to(torch.float32).contiguous()
            keep.append(t)
            setattr(gr, name, t.data_ptr())
    assert param_grads.dtype == torch.float32 and param_grads.is_contiguous() and param_grads.numel() == (NUM_PARAMS_EXT if cfg.ipe_groups > 1 else NUM_PARAMS)
    ws = backward_workspace(R, N, param_grads.device, cfg.ipe_groups)
    check(lib().refnerf_level_backward(ptr(packed), C.byref(cfg), C.byref(rs), R, C.byref(sv), C.byref(gr),
                                       ptr(param_grads), ptr(ws), ws.numel(), stream_ptr()))
    return param_grads


def render_rays(cfg: LevelCfg, density, tdist, directions, far, rgb=None, diffuse=None, specular=None, normals=None,
                normals_pred=None, roughness=None, tint=None):
    """render.compute_alpha_weights + render.volumetric_rendering (render.py:132-254) on caller-supplied per-sample
    device tensors: density / roughness [R,N], tdist [R,N+1], directions [R,3], far [R], the others [R,N,3] or None.
    cfg.n_samples is taken from `density`.  Returns dict(weights, r_rgb, ..., r_percentiles)."""
    require_device()
    dev = density.device
    R, N = density.shape
    cfg.n_samples = N
    f32 = dict(dtype=torch.float32, device=dev)
    keep = []

    def prep(t, shape):
        if t is None:
            return None
        t = t.to(torch.float32).reshape(shape).contiguous()
        keep.append(t)
        return C.c_void_p(t.data_ptr())
    args = [prep(density, (R, N)), prep(tdist, (R, N + 1)), prep(directions, (R, 3)), prep(far, (R,)),
            prep(rgb, (R, N, 3)), prep(diffuse, (R, N, 3)), prep(specular, (R, N, 3)), prep(normals, (R, N, 3)),
            prep(normals_pred, (R, N, 3)), prep(roughness, (R, N)), prep(tint, (R, N, 3))]
    res = {"weights": torch.empty((R, N), **f32), "r_rgb": torch.empty((R, 3), **f32),
           "r_diffuse": torch.empty((R, 3), **f32), "r_specular": torch.empty((R, 3), **f32),
           "r_distance": torch.empty((R,), **f32), "r_acc": torch.empty((R,), **f32)}
    if cfg.compute_extras:
        res.update({"r_normals_pred": torch.empty((R, 3), **f32), "r_tint": torch.empty((R, 3), **f32),
                    "r_roughness": torch.empty((R,), **f32), "r_distance_mean": torch.empty((R,), **f32),
                    "r_percentiles": torch.empty((R, 3), dtype=torch.float64, device=dev)})
        if normals is not None:
            res["r_normals"] = torch.empty((R, 3), **f32)
    out = LevelOut()
    for k, t in res.items():
        setattr(out, "d_" + k, t.data_ptr())
    check(lib().refnerf_render_rays(C.byref(cfg), R, *args, C.byref(out), stream_ptr()))
    return res


def losses_forward(r_rgb, gt_rgb, lossmult, weights, orientation_normals, normals, normals_pred, viewdirs):
    """refnerf_losses_forward: per-ray terms [R,3] of the three Ref-NeRF losses of one level (device tensors; the
    normals arguments may be None = term off)."""
    require_device()
    R, N = weights.shape
    terms = torch.empty((R, 3), dtype=torch.float32, device=weights.device)
    check(lib().refnerf_losses_forward(R, N, ptr(r_rgb), ptr(gt_rgb), ptr(lossmult), ptr(weights), ptr(orientation_normals),
                                       ptr(normals), ptr(normals_pred), ptr(viewdirs), ptr(terms), stream_ptr()))
    return terms


def losses_backward(r_rgb, gt_rgb, lossmult, weights, orientation_normals, orientation_on_pred, normals, normals_pred,
                    viewdirs, g_data, g_orientation, g_normal, upstream=None):
    """refnerf_losses_backward -> (g_r_rgb [R,3], g_weights [R,N], g_normals_pred [R,N,3])."""
    require_device()
    R, N = weights.shape
    f32 = dict(dtype=torch.float32, device=weights.device)
    g_rgb, g_w, g_np = torch.empty((R, 3), **f32), torch.empty((R, N), **f32), torch.empty((R, N, 3), **f32)
    check(lib().refnerf_losses_backward(R, N, ptr(r_rgb), ptr(gt_rgb), ptr(lossmult), ptr(weights), ptr(orientation_normals),
                                        int(orientation_on_pred), ptr(normals), ptr(normals_pred), ptr(viewdirs),
                                        float(g_data), float(g_orientation), float(g_normal), ptr(upstream),
                                        ptr(g_rgb), ptr(g_w), ptr(g_np), stream_ptr()))
    return g_rgb, g_w, g_np


def sample_intervals(t, logits, n, smin=0.0, smax=1.0):
    require_device()
    R, M = logits.shape
    sd = torch.empty((R, n + 1), dtype=torch.float32, device=t.device)
    bi = torch.empty((R, n), dtype=torch.int32, device=t.device)
    check(lib().refnerf_sample_intervals(ptr(t.contiguous()), ptr(logits.contiguous()), R, M, n, smin, smax,
                                         ptr(sd), ptr(bi), stream_ptr()))
    return sd, bi


def integrated_pos_enc(lmean, lvar):
    require_device()
    n = lmean.numel() // 3
    out = torch.empty(lmean.shape[:-1] + (96,), dtype=torch.float32, device=lmean.device)
    check(lib().refnerf_integrated_pos_enc(ptr(lmean.contiguous()), ptr(lvar.contiguous()), n, ptr(out), stream_ptr()))
    return out


def integrated_dir_enc(xyz, kappa_inv):
    require_device()
    n = xyz.numel() // 3
    out = torch.empty(xyz.shape[:-1] + (72,), dtype=torch.float32, device=xyz.device)
    k = kappa_inv.to(torch.float32).expand(xyz.shape[:-1] + (1,)).contiguous()
    check(lib().refnerf_integrated_dir_enc(ptr(xyz.contiguous()), ptr(k), n, ptr(out), stream_ptr()))
    return out


def set_timing(enable: bool):
    lib().refnerf_set_timing(int(enable))


TIMER_FORWARD, TIMER_BACKWARD, TIMER_WGRAD = 0, 1, 2


def get_timing(family: int = TIMER_FORWARD):
    """(total ms, launches) of the kernels of one family launched since set_timing(True)."""
    ms, n = C.c_double(0), C.c_int64(0)
    check(lib().refnerf_get_timing_family(family, C.byref(ms), C.byref(n)))
    return ms.value, n.value
