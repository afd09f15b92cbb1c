"""Deterministic synthetic rays / weights / targets.

No datasets or checkpoints exist in the build or GPU containers, so tests,
golden-vector capture and bench.py all draw their inputs from here.  The
generator is a counter-based hash (splitmix64 finaliser) so that the same
numbers can be reproduced anywhere without depending on a library RNG stream.

Ray generation restates the pinhole / NDC maths of the reference's
``camera_utils.pixels_to_rays`` (camera_utils.py:502-614) and
``convert_to_ndc`` (camera_utils.py:31-97) for the two synthetic families:
Blender-style 800x800 f=1111.1 at distance 4 (near 2 / far 6) and LLFF-style
1008x756 f=815 NDC (near 0 / far 1).
"""
import math

import numpy as np

from . import layout

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def hash_uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n float64 values in [0, 1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64)
        x = x + np.uint64((seed * 0x9E3779B97F4A7C15 + stream * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def make_params(seed: int = 0, bias_scale: float = 0.0, sharpen: float = 1.0,
                roughness_bias: float = 0.0) -> np.ndarray:
    """Canonical parameter blob (layout.PARAM_SPECS order), float32.

    Weights ~ U(+-1/sqrt(fan_in)) and zero bias is the reference's init
    (models.py:38-47).  ``bias_scale`` > 0 draws biases U(+-bias_scale) so tests
    exercise the bias path; ``sharpen`` multiplies raw_density.weight (peaky
    weights -> non-uniform level-1 resampling); ``roughness_bias`` sets
    raw_roughness.bias (-6 ~ roughness 1e-3, the "shiny" case).
    """
    blob = np.zeros(layout.NUM_PARAMS, dtype=np.float32)
    for sid, spec in enumerate(layout.PARAM_SPECS):
        n = spec.out_dim * spec.in_dim
        bound = 1.0 / math.sqrt(spec.in_dim)
        w = (hash_uniform(seed, 2 * sid, n) * 2.0 - 1.0) * bound
        if spec.name == "raw_density":
            w = w * sharpen
        blob[spec.w_off:spec.w_off + n] = w.astype(np.float32)
        if bias_scale > 0.0:
            b = (hash_uniform(seed, 2 * sid + 1, spec.out_dim) * 2.0 - 1.0) * bias_scale
            blob[spec.b_off:spec.b_off + spec.out_dim] = b.astype(np.float32)
        if spec.name == "raw_roughness" and roughness_bias != 0.0:
            blob[spec.b_off] = np.float32(roughness_bias)
    return blob


def make_basis_params(seed: int = 0, n_basis: int = 21, **kw) -> np.ndarray:
    """Extended canonical blob (layout.NUM_PARAMS_EXT) for an MLP with a general IPE basis of `n_basis` directions:
    make_params(seed, **kw) with the IPE columns of spatial_net.0 / .5 re-drawn for the wider fan-in, plus the tail
    blocks of direction groups 1 .. n_basis / 3 - 1 (unused groups stay zero)."""
    blob = np.zeros(layout.NUM_PARAMS_EXT, dtype=np.float32)
    blob[:layout.NUM_PARAMS] = make_params(seed=seed, **kw)
    specs, idx = layout.variant_layout(n_basis=n_basis)
    for sid, name in ((91, "spatial_net.0"), (92, f"spatial_net.{layout.SKIP + 1}")):
        sp = next(s for s in specs if s.name == name)
        n = sp.out_dim * sp.in_dim
        w = ((hash_uniform(seed, sid, n) * 2.0 - 1.0) / math.sqrt(sp.in_dim)).astype(np.float32)
        blob[idx[sp.w_off:sp.w_off + n]] = w
    return blob


def _rot(seed: int) -> np.ndarray:
    """Seeded rotation matrix (float64)."""
    u = hash_uniform(seed, 101, 3)
    q = np.array([math.sqrt(1 - u[0]) * math.sin(2 * math.pi * u[1]),
                  math.sqrt(1 - u[0]) * math.cos(2 * math.pi * u[1]),
                  math.sqrt(u[0]) * math.sin(2 * math.pi * u[2]),
                  math.sqrt(u[0]) * math.cos(2 * math.pi * u[2])])
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _pixels_to_rays(px, py, focal, width, height, c2w, ndc_near=None):
    """Pinhole rays through pixel centres; restates camera_utils.py:548-612."""
    px = px.astype(np.float32)
    py = py.astype(np.float32)
    pixtocam = np.linalg.inv(np.array([[focal, 0, width / 2.0],
                                       [0, focal, height / 2.0],
                                       [0, 0, 1.0]])).astype(np.float32)

    def to_dir(x, y):
        p = np.stack([x + 0.5, y + 0.5, np.ones_like(x)], -1)
        cam = (pixtocam @ p[..., None])[..., 0]
        cam = cam @ np.diag(np.array([1.0, -1.0, -1.0], dtype=np.float32))
        return (c2w[:3, :3].astype(np.float32) @ cam[..., None])[..., 0], cam

    d, cam = to_dir(px, py)
    dx, _ = to_dir(px + 1, py)
    dy, _ = to_dir(px, py + 1)
    o = np.broadcast_to(c2w[:3, 3].astype(np.float32), d.shape).copy()
    viewdirs = d / np.linalg.norm(d, axis=-1, keepdims=True)
    if ndc_near is None:
        dxn = np.linalg.norm(dx - d, axis=-1)
        dyn = np.linalg.norm(dy - d, axis=-1)
    else:
        def ndc(o_, d_):
            t = -(ndc_near + o_[..., 2]) / d_[..., 2]
            o2 = o_ + t[..., None] * d_
            xm = 1.0 / pixtocam[0, 2]
            ym = 1.0 / pixtocam[1, 2]
            on = np.stack([xm * o2[..., 0] / o2[..., 2], ym * o2[..., 1] / o2[..., 2], -np.ones_like(t)], -1)
            inf = np.stack([xm * d_[..., 0] / d_[..., 2], ym * d_[..., 1] / d_[..., 2], np.ones_like(t)], -1)
            return on, inf - on
        ox, _ = ndc(o, dx)
        oy, _ = ndc(o, dy)
        o, d = ndc(o, d)
        dxn = np.linalg.norm(ox - o, axis=-1)
        dyn = np.linalg.norm(oy - o, axis=-1)
    radii = (0.5 * (dxn + dyn))[..., None] * 2 / np.sqrt(12)
    return (o.astype(np.float32), d.astype(np.float32), viewdirs.astype(np.float32),
            radii.astype(np.float32), cam[..., :2].astype(np.float32))


def _bundle(o, d, v, radii, imageplane, near, far):
    n = o.shape[0]
    return dict(origins=o, directions=d, viewdirs=v, radii=radii, imageplane=imageplane,
                lossmult=np.ones((n, 1), np.float32),
                near=np.full((n, 1), near, np.float32), far=np.full((n, 1), far, np.float32),
                cam_idx=np.zeros((n, 1), np.float32))


def blender_camera(seed: int = 1, width: int = 800):
    """(camtoworld [3,4] float64, focal) of the synthetic Blender-style view used by blender_rays."""
    focal = 0.5 * 800 / math.tan(0.5 * 0.6911112070083618) * (width / 800.0)
    rot = _rot(seed)
    c2w = np.zeros((3, 4))
    c2w[:3, :3] = rot
    c2w[:3, 3] = rot @ np.array([0.0, 0.0, 4.0])
    return c2w, focal


def blender_rays(n_rays: int, seed: int = 1, full_image: bool = False,
                 width: int = 800, height: int = 800, center_frac: float = 1.0) -> dict:
    """Blender-style rays: pinhole, camera at distance 4 looking at the origin."""
    focal = 0.5 * 800 / math.tan(0.5 * 0.6911112070083618) * (width / 800.0)
    rot = _rot(seed)
    c2w = np.zeros((3, 4))
    c2w[:3, :3] = rot
    c2w[:3, 3] = rot @ np.array([0.0, 0.0, 4.0])
    if full_image:
        yy, xx = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
        px, py = xx.reshape(-1), yy.reshape(-1)
    else:
        lo_x = 0.5 * (1 - center_frac) * width
        lo_y = 0.5 * (1 - center_frac) * height
        px = np.floor(lo_x + hash_uniform(seed, 1, n_rays) * center_frac * width).astype(np.int64)
        py = np.floor(lo_y + hash_uniform(seed, 2, n_rays) * center_frac * height).astype(np.int64)
    return _bundle(*_pixels_to_rays(px, py, focal, width, height, c2w), near=2.0, far=6.0)


def llff_rays(n_rays: int, seed: int = 1, width: int = 1008, height: int = 756, full_image: bool = False) -> dict:
    """LLFF-style forward-facing rays in NDC (near 0 / far 1)."""
    focal = 815.0
    u = hash_uniform(seed, 7, 3)
    c2w = np.zeros((3, 4))
    c2w[:3, :3] = np.eye(3)
    c2w[:3, 3] = (u - 0.5) * np.array([0.6, 0.4, 0.2])
    if full_image:
        yy, xx = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
        px, py = xx.reshape(-1), yy.reshape(-1)
    else:
        px = np.floor(hash_uniform(seed, 1, n_rays) * width).astype(np.int64)
        py = np.floor(hash_uniform(seed, 2, n_rays) * height).astype(np.int64)
    return _bundle(*_pixels_to_rays(px, py, focal, width, height, c2w, ndc_near=1.0), near=0.0, far=1.0)


def target_rgb(n_rays: int, seed: int = 2) -> np.ndarray:
    return hash_uniform(seed, 3, n_rays * 3).reshape(n_rays, 3).astype(np.float32)
