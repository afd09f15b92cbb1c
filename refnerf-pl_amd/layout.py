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
