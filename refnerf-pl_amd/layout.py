"""Canonical parameter layout of the Ref-NeRF NerfMLP.

One flat float32 blob in the reference's ``state_dict`` order for
``nerf_mlp.*`` (models.py:497-531): each tensor row-major ``[out][in]``, weight
then bias.  This is the layout the C ABI (include/refnerf_hip.h) and the oracle
(oracle/refnerf_oracle.h, ``rn_param_layout``) both take.
"""
from dataclasses import dataclass
from typing import List

WIDTH = 256      # NerfMLP.net_width / net_width_viewdirs
DEPTH = 8        # NerfMLP.net_depth / net_depth_viewdirs
SKIP = 4         # MLP.skip_layer
IPE_DIM = 96     # 2 * max_deg_point(16) * 3
BNECK = 128
IDE_DIM = 72
DIR_IN = BNECK + IDE_DIM + 1


@dataclass(frozen=True)
class ParamSpec:
    name: str        # module path under nerf_mlp, e.g. "spatial_net.3"
    out_dim: int
    in_dim: int
    w_off: int
    b_off: int


def _build() -> List[ParamSpec]:
    specs, p = [], 0

    def add(name, out_dim, in_dim):
        nonlocal p
        specs.append(ParamSpec(name, out_dim, in_dim, p, p + out_dim * in_dim))
        p += out_dim * in_dim + out_dim

    for i in range(DEPTH):
        add(f"spatial_net.{i}", WIDTH, IPE_DIM if i == 0 else (WIDTH + IPE_DIM if i == SKIP + 1 else WIDTH))
    add("raw_density", 1, WIDTH)
    add("grad_pred", 3, WIDTH)
    add("raw_roughness", 1, WIDTH)
    add("raw_rgb_diffuse", 3, WIDTH)
    add("raw_tint", 3, WIDTH)
    add("bottleneck", BNECK, WIDTH)
    for i in range(DEPTH):
        add(f"viewdir_mlp.{i}", WIDTH, DIR_IN if i == 0 else (WIDTH + DIR_IN if i == SKIP + 1 else WIDTH))
    add("rgb", 3, WIDTH)
    return specs


PARAM_SPECS: List[ParamSpec] = _build()
NUM_PARAMS: int = PARAM_SPECS[-1].b_off + PARAM_SPECS[-1].out_dim
assert NUM_PARAMS == 1110158
SPEC_BY_NAME = {s.name: s for s in PARAM_SPECS}


# ---------------------------------------------------------------------------------------------------------------------
# Variants of the NerfMLP that run in the reference and still fit the fused kernels' topology (SURVEY section 8 row f4;
# tests/golden/variants_status.json records which flag settings the reference itself survives).  They are served by
# EMBEDDING their parameters into the canonical blob above -- the kernels and the C ABI see the Ref-NeRF network:
#   net_width_viewdirs = Wv <= 256   rows / hidden columns >= Wv of the directional layers are zero (dead units)
#   use_n_dot_v = False              the n.v column of the two layers that read the dir input is zero
#   use_specular_tint = False        raw_tint absent: zero weights and bias -> tint = sigmoid(0) = 0.5 exactly, i.e.
#                                    specular = 0.5 rgb (internal/models.py:708-709)
#   enable_pred_roughness = False    raw_roughness absent: zero weights, and the level runs with roughness_bias = -inf-ish
#                                    (ROUGHNESS_OFF_BIAS): softplus -> exactly 0, the IDE at zero roughness (models.py:636-641)
#   use_directional_enc = False      coord.pos_enc of the reflected direction (models.py:487-492): the kernels compute it
#                                    INTO the IDE's 72 slots (cfg.dir_enc = REFNERF_DIRENC_POSENC), its 3 + 6 deg_view weight
#                                    columns sit at posenc_slots()
# The module keeps the reference's parameter names and TRUE shapes (checkpoints load unchanged); `variant_layout` gives the
# true-shape spec list and, for every true parameter element, its position in the canonical blob.
ROUGHNESS_OFF_BIAS = -1.0e30


POSENC_MAX_DEG = 5   # degrees of coord.pos_enc the directional slots hold (3 + 15 + 15 features)


def posenc_slots(deg_view):
    """Slot (0..71, relative to the bottleneck's end) of every feature of coord.pos_enc(d, 0, deg_view, append_identity=True)
    in the kernels' directional encoding block: [x y z | sin(2^j x_i) (j-major) | 0.. || sin(2^j x_i + pi/2) | 0..]."""
    if not 1 <= int(deg_view) <= POSENC_MAX_DEG:
        raise ValueError(f"pos_enc view encoding: deg_view must be in [1, {POSENC_MAX_DEG}] for the fused kernels")
    n = 3 * int(deg_view)
    return list(range(3)) + [3 + k for k in range(n)] + [IDE_DIM // 2 + k for k in range(n)]


# A general IPE basis (NerfMLP.basis_shape / basis_subdivisions: n_basis = 3 G directions, G <= 7) widens spatial_net.0 / .5
# to 32 n_basis IPE columns.  The kernels take them as G groups of three directions (csrc/refnerf_layout.h): group 0 in the
# canonical blob's 96 IPE columns, groups 1.. in a TAIL behind the canonical blob, W_ext[layer 0 | 5][256][96 (g - 1) + k]; column
# 16 n c + n j + d of the true weight (cos block c, degree j, direction d; coord.py:102-126) = column 48 c + 3 j + d % 3 of
# group d // 3.
IPE_MAX_GROUPS = 7
EXT_GROUPS = IPE_MAX_GROUPS - 1
NUM_PARAMS_EXT = NUM_PARAMS + 2 * EXT_GROUPS * WIDTH * IPE_DIM


def ipe_column_positions(spec, layer_slot, col0, n_basis, rows=None, min_deg=0, max_deg=IPE_DIM // 6):
    """canonical / tail positions of the 2 (max_deg - min_deg) n_basis IPE columns of the rows `rows` (default: all 256) of
    spatial_net.0 (layer_slot 0, col0 0) or spatial_net.5 (layer_slot 1, col0 256): int64 [len(rows), columns].  The
    kernels always evaluate degrees 0..15 (scale 2^j); the columns of absent degrees keep zero weights."""
    import numpy as np
    c, j, d = np.meshgrid(np.arange(2), np.arange(min_deg, max_deg), np.arange(n_basis), indexing="ij")
    g, k = d // 3, (IPE_DIM // 2) * c + 3 * j + d % 3
    rows = np.arange(WIDTH, dtype=np.int64)[:, None] if rows is None else np.asarray(rows, np.int64)[:, None]
    in_canon = spec.w_off + rows * spec.in_dim + (col0 + k.reshape(-1))[None, :]
    in_tail = NUM_PARAMS + (layer_slot * WIDTH + rows) * (EXT_GROUPS * IPE_DIM) + ((g.reshape(-1) - 1) * IPE_DIM + k.reshape(-1))[None, :]
    return np.where(g.reshape(-1)[None, :] == 0, in_canon, in_tail)


def _check_depth(name, d):
    d = int(d)
    if not 1 <= d <= DEPTH or d == SKIP + 1:
        raise ValueError(f"{name} must be in [1, {DEPTH}] and not {SKIP + 1} for the fused kernels (with {SKIP + 1} layers the reference "
                         f"feeds the skip concatenation straight into the heads), got {d}")
    return d


def identity_fill(net_depth=DEPTH, net_depth_viewdirs=DEPTH, net_width=WIDTH, net_width_viewdirs=WIDTH):
    """Shallower trunks (net_depth / net_depth_viewdirs < 8) run as the canonical 8 layers with IDENTITY layers behind the
    real ones: the activations are post-ReLU (>= 0), so relu(1 x + 0) = x -- exactly in the f32, bf16 and f16 modes (x is
    already a value of the operand type); in the split-f16 modes each identity layer re-rounds x to its hi + lo pair (22
    bits: <= 2^-22 relative per layer) -- and the backward passes the deltas through unchanged.  -> int64 positions of the canonical blob that hold 1.0 (not parameters)."""
    import numpy as np
    out = []
    for pre, d, w in (("spatial_net.", _check_depth("net_depth", net_depth), int(net_width)),
                      ("viewdir_mlp.", _check_depth("net_depth_viewdirs", net_depth_viewdirs), int(net_width_viewdirs))):
        for i in range(d, DEPTH):
            c = SPEC_BY_NAME[pre + str(i)]
            u = np.arange(w, dtype=np.int64)
            out.append(c.w_off + u * c.in_dim + u)
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def variant_layout(net_width_viewdirs=WIDTH, use_n_dot_v=True, use_specular_tint=True, enable_pred_roughness=True,
                   use_directional_enc=True, deg_view=5, n_basis=3, net_width=WIDTH, bottleneck_width=BNECK,
                   min_deg_point=0, max_deg_point=IPE_DIM // 6, net_depth=DEPTH, net_depth_viewdirs=DEPTH):
    """-> (specs, index): `specs` = ParamSpec list of the variant (true shapes, offsets into ITS flat blob, state_dict
    order), `index` = int64 numpy array, index[i] = canonical-blob position of element i of the variant's flat blob; or
    (PARAM_SPECS, None) for the Ref-NeRF network itself.
    Narrower networks (net_width / net_width_viewdirs <= 256, bottleneck_width <= 128) are dead units of the canonical one
    (zero rows and columns: exact), fewer IPE degrees (0 <= min_deg_point < max_deg_point <= 16) zero columns, shallower
    trunks (net_depth / net_depth_viewdirs < 8) leave out the layers that identity_fill() turns into identities."""
    import numpy as np
    wv, w, bw = int(net_width_viewdirs), int(net_width), int(bottleneck_width)
    lo, hi = int(min_deg_point), int(max_deg_point)
    if not 1 <= wv <= WIDTH:
        raise ValueError(f"net_width_viewdirs must be in [1, {WIDTH}] for the fused kernels, got {wv}")
    if not 1 <= w <= WIDTH:
        raise ValueError(f"net_width must be in [1, {WIDTH}] for the fused kernels, got {w}")
    if not 1 <= bw <= BNECK:
        raise ValueError(f"bottleneck_width must be in [1, {BNECK}] for the fused kernels, got {bw}")
    if not 0 <= lo < hi <= IPE_DIM // 6:
        raise ValueError(f"min_deg_point / max_deg_point must satisfy 0 <= min < max <= {IPE_DIM // 6} for the fused kernels, got {lo}, {hi}")
    if n_basis % 3 or not 3 <= n_basis <= 3 * IPE_MAX_GROUPS:
        raise ValueError(f"IPE basis of {n_basis} directions: the fused kernels take 3, 6, ... {3 * IPE_MAX_GROUPS} "
                         "(octahedron / 1-2, icosahedron / 1-2)")
    ds, dv = _check_depth("net_depth", net_depth), _check_depth("net_depth_viewdirs", net_depth_viewdirs)
    if (wv == WIDTH and w == WIDTH and bw == BNECK and (lo, hi) == (0, IPE_DIM // 6) and use_n_dot_v and use_specular_tint
            and enable_pred_roughness and use_directional_enc and n_basis == 3 and ds == DEPTH and dv == DEPTH):
        return PARAM_SPECS, None
    enc_cols = list(range(IDE_DIM)) if use_directional_enc else posenc_slots(deg_view)
    din_cols = list(range(bw)) + [BNECK + k for k in enc_cols] + ([BNECK + IDE_DIM] if use_n_dot_v else [])
    specs, idx, p = [], [], 0
    for c in PARAM_SPECS:
        if c.name == "raw_tint" and not use_specular_tint:
            continue
        if c.name == "raw_roughness" and not enable_pred_roughness:
            continue
        rows = list(range(c.out_dim))
        cols = list(range(c.in_dim))
        ipe = None                                   # (layer slot, first canonical IPE column) of the two layers that read the IPE
        if c.name.startswith("spatial_net."):
            i = int(c.name.split(".")[1])
            if i >= ds:
                continue                             # an identity layer of the canonical network (identity_fill)
            rows = list(range(w))
            cols = [] if i == 0 else list(range(w))
            if i == 0 or i == SKIP + 1:
                ipe = (0, 0) if i == 0 else (1, WIDTH)
        elif c.name.startswith("viewdir_mlp."):
            i = int(c.name.split(".")[1])
            if i >= dv:
                continue
            rows = list(range(wv))
            if i == 0:
                cols = din_cols
            elif i == SKIP + 1:
                cols = list(range(wv)) + [WIDTH + k for k in din_cols]
            else:
                cols = list(range(wv))
        elif c.name == "rgb":
            cols = list(range(wv))
        else:                                        # heads on the spatial trunk's output
            cols = list(range(w))
            if c.name == "bottleneck":
                rows = list(range(bw))
        r = np.asarray(rows, np.int64)[:, None]
        k = np.asarray(cols, np.int64)[None, :]
        pos = c.w_off + r * c.in_dim + k
        if ipe is not None:
            pos = np.concatenate([pos, ipe_column_positions(c, ipe[0], ipe[1], n_basis, rows, lo, hi)], axis=1)
        out_dim, in_dim = pos.shape
        specs.append(ParamSpec(c.name, out_dim, in_dim, p, p + out_dim * in_dim))
        p += out_dim * in_dim + out_dim
        idx.append(pos.reshape(-1))
        idx.append(c.b_off + np.asarray(rows, np.int64))
    return specs, np.concatenate(idx)
