"""Host-side mirror of the reference's ``internal/models.py`` call surface.

``Model``, ``MLP``, ``NerfMLP``, ``PropMLP``, ``construct_model`` and
``render_image`` keep the reference's names, constructor parameters, gin
names, ``state_dict`` keys and return contract, but the level
loop body of ``Model.__call__`` (models.py:162-306) is ONE fused HIP launch per
level through the C ABI of include/refnerf_hip.h.  There is no PyTorch / CPU
fallback: without the library or a gfx950 device the call raises.
"""
import contextlib
import math as python_math
import threading
from typing import Any, Callable, List, Mapping, MutableMapping, Optional, Text, Tuple

import numpy as np
import torch
from torch import nn

from . import _hip, configs, geopoly, layout, utils


def _reset_parameters(linear: nn.Linear):
    """models.py:38-47: U(+-1/sqrt(fan_in)) weights, zero bias."""
    nn.init.kaiming_uniform_(linear.weight, a=python_math.sqrt(5))
    if linear.bias is not None:
        nn.init.constant_(linear.bias, val=0)


def _linear(in_f, out_f):
    lin = nn.Linear(in_f, out_f)
    _reset_parameters(lin)
    return lin


class MLP(nn.Module):
    """A PosEnc MLP (models.py:343-750): parameter container + configuration.

    Only the Ref-NeRF architecture family of the shipped refnerf configs is built
    as a fused kernel; other flag combinations raise ``ValueError`` at
    construction (they are valid in the reference).
    """

    def __init__(
            self,
            net_depth: int = 8,
            net_width: int = 256,
            bottleneck_width: int = 256,
            net_depth_viewdirs: int = 1,
            net_width_viewdirs: int = 128,
            net_activation: Callable[..., Any] = torch.nn.functional.relu,
            min_deg_point: int = 0,
            max_deg_point: int = 12,
            weight_init: str = 'he_uniform',
            skip_layer: int = 4,
            skip_layer_dir: int = 4,
            num_rgb_channels: int = 3,
            deg_view: int = 4,
            use_reflections: bool = False,
            use_directional_enc: bool = False,
            enable_pred_roughness: bool = False,
            roughness_activation: Callable[..., Any] = torch.nn.functional.softplus,
            roughness_bias: float = -1.,
            use_diffuse_color: bool = False,
            use_specular_tint: bool = False,
            use_n_dot_v: bool = False,
            enable_pred_specular_density: bool = False,
            bottleneck_noise: float = 0.0,
            density_activation: Callable[..., Any] = torch.nn.functional.softplus,
            density_bias: float = -1.,
            density_noise: float = 0.,
            rgb_premultiplier: float = 1.,
            rgb_activation: Callable[..., Any] = torch.sigmoid,
            rgb_bias: float = 0.,
            rgb_padding: float = 0.001,
            enable_pred_normals: bool = False,
            disable_density_normals: bool = False,
            disable_rgb: bool = False,
            srgb_mapping: bool = True,
            srgb_mapping_normalization: bool = True,
            warp_fn: Callable[..., Any] = None,
            basis_shape: str = 'icosahedron',
            basis_subdivisions: int = 2,
    ):
        super().__init__()
        for k, v in list(locals().items()):
            if k not in ("self", "__class__"):
                setattr(self, k, v)

        # models.py:471-480 (same errors as the reference)
        if self.use_reflections and not (self.enable_pred_normals or not self.disable_density_normals):
            raise ValueError('Normals must be computed for reflection directions.')
        if self.enable_pred_specular_density and not self.use_diffuse_color:
            raise ValueError('Specular density is useless if not using diffuse color.')
        self._check_supported()

        # the IPE basis (models.py:482-484): 'octahedron' / 1 = the three directions the kernels are built around; other
        # tesselations (the constructor default 'icosahedron' / 2 has 21 directions) go through the kernels as groups of
        # three (csrc/refnerf_layout.h), f32 mode
        basis = geopoly.generate_basis(self.basis_shape, self.basis_subdivisions)
        self.pos_basis_t = torch.tensor(basis).T                      # [3, n] as in the reference
        self.ipe_basis_dirs = int(basis.shape[0])
        self._basis_np = np.ascontiguousarray(basis, np.float32)
        self._basis_dev = None
        # same module names / shapes as the reference after its lazy init (models.py:497-531)
        W = self.net_width
        ipe_dim = 2 * (self.max_deg_point - self.min_deg_point) * self.ipe_basis_dirs
        sp_in = [ipe_dim if i == 0 else (W + ipe_dim if i == self.skip_layer + 1 else W)
                 for i in range(self.net_depth)]
        self.spatial_net = nn.ModuleList([_linear(sp_in[i], W) for i in range(self.net_depth)])
        self.raw_density = _linear(W, 1)
        if self.enable_pred_specular_density:                  # models.py:502-503 (same place in the construction order)
            self.raw_specular_density = _linear(W, 1)
        self.grad_pred = _linear(W, 3)
        # the variants the reference runs and the fused kernels serve by embedding (layout.variant_layout): same module
        # names and TRUE shapes as the reference gives them (models.py:509-531)
        self.specs, idx = layout.variant_layout(self.net_width_viewdirs, self.use_n_dot_v, self.use_specular_tint,
                                                self.enable_pred_roughness, self.use_directional_enc, self.deg_view,
                                                n_basis=self.ipe_basis_dirs, net_width=self.net_width,
                                                bottleneck_width=self.bottleneck_width, min_deg_point=self.min_deg_point,
                                                max_deg_point=self.max_deg_point, net_depth=self.net_depth,
                                                net_depth_viewdirs=self.net_depth_viewdirs)
        self._identity_fill_np = layout.identity_fill(self.net_depth, self.net_depth_viewdirs, self.net_width, self.net_width_viewdirs)
        self.canon_size = layout.NUM_PARAMS_EXT if self.ipe_basis_dirs != 3 else layout.NUM_PARAMS
        self.num_params = self.specs[-1].b_off + self.specs[-1].out_dim
        self._embed_index_np = idx                 # None: the parameters ARE the canonical blob
        self._embed_index = None
        self._canon = None
        if self.enable_pred_roughness:
            self.raw_roughness = _linear(W, 1)
        self.raw_rgb_diffuse = _linear(W, self.num_rgb_channels)
        if self.use_specular_tint:
            self.raw_tint = _linear(W, 3)
        self.bottleneck = _linear(W, self.bottleneck_width)
        Wv = self.net_width_viewdirs
        enc = layout.IDE_DIM if self.use_directional_enc else 3 + 6 * self.deg_view      # IDE / coord.pos_enc (models.py:484-492)
        din = self.bottleneck_width + enc + (1 if self.use_n_dot_v else 0)
        vd_in = [din if i == 0 else (Wv + din if i == self.skip_layer + 1 else Wv)
                 for i in range(self.net_depth_viewdirs)]
        self.viewdir_mlp = nn.ModuleList([_linear(vd_in[i], Wv) for i in range(self.net_depth_viewdirs)])
        self.rgb = _linear(Wv, self.num_rgb_channels)

        self._flat = None          # flat blob (state_dict order, true shapes) the parameters alias
        self._packed = None        # MFMA operand image
        self._packed_key = None

    # ---- supported-configuration gate ------------------------------------
    def _check_supported(self):
        # Flag settings the REFERENCE does not survive (tests/golden/variants_status.json, captured from it): its
        # Model.__call__ reads ray_results['diffuse'] unconditionally (models.py:272), reflect() needs predicted normals
        # in eval mode (ref_utils.py:37), and the un-reflected encoding is broadcast with one dimension too many
        # (models.py:671).  That includes both shipped mip-NeRF configs; the same settings raise here.
        dead = {}
        if not self.use_diffuse_color:
            dead["use_diffuse_color"] = "False: KeyError 'diffuse' at internal/models.py:272 in the reference"
        if not self.use_reflections:
            dead["use_reflections"] = "False: RuntimeError in torch.broadcast_to at internal/models.py:671 in the reference"
        if not self.enable_pred_normals:
            dead["enable_pred_normals"] = "False: TypeError in ref_utils.reflect at internal/ref_utils.py:37 in the reference"
        if dead:
            raise ValueError(f"MLP flags the reference itself cannot run (so there is nothing to match): {dead}")
        want = dict(skip_layer=4, num_rgb_channels=3, bottleneck_noise=0.0,
                    density_noise=0., disable_rgb=False, warp_fn=None)
        bad = {k: getattr(self, k) for k, v in want.items() if getattr(self, k) != v}
        for k, top in (("net_width_viewdirs", layout.WIDTH), ("net_width", layout.WIDTH), ("bottleneck_width", layout.BNECK)):
            if not 1 <= int(getattr(self, k)) <= top:
                bad[k] = getattr(self, k)
        for k in ("net_depth", "net_depth_viewdirs"):
            if not 1 <= int(getattr(self, k)) <= layout.DEPTH or int(getattr(self, k)) == layout.SKIP + 1:
                bad[k] = getattr(self, k)
        if not 0 <= int(self.min_deg_point) < int(self.max_deg_point) <= layout.IPE_DIM // 6:
            bad["min_deg_point / max_deg_point"] = (self.min_deg_point, self.max_deg_point)
        if (self.use_directional_enc and self.deg_view != 5) or not 1 <= int(self.deg_view) <= layout.POSENC_MAX_DEG:
            bad["deg_view"] = self.deg_view            # the IDE table is built for degree 5, pos_enc has slots for <= 5
        if bad:
            raise ValueError(
                "MLP configuration outside the fused Ref-NeRF family (configs/*refnerf*.gin; served variants: "
                "net_width / net_width_viewdirs <= 256, bottleneck_width <= 128, net_depth / net_depth_viewdirs in 1..8 except 5, IPE degrees within [0, 16], bases of <= 21 directions, "
                "use_n_dot_v / use_specular_tint / enable_pred_roughness / "
                f"use_directional_enc / disable_density_normals either way): {bad}; expected {({k: want.get(k, 'see above') for k in bad})}")
        if self.net_activation is not torch.nn.functional.relu:
            raise ValueError("net_activation must be relu")
        if self.density_activation is not torch.nn.functional.softplus or \
                self.roughness_activation is not torch.nn.functional.softplus or \
                self.rgb_activation is not torch.sigmoid:
            raise ValueError("density/roughness activations must be softplus and rgb_activation sigmoid")

    # ---- flat canonical blob ----------------------------------------------
    def _named_linears(self):
        for spec in self.specs:
            mod = self
            for part in spec.name.split("."):
                mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
            yield spec, mod

    def flat_params(self) -> torch.Tensor:
        """The 46 parameters as ONE canonical fp32 blob; the nn.Parameters are
        re-pointed to views of it (so optimiser steps update the blob in place)."""
        first = self.spatial_net[0].weight
        ok = (self._flat is not None and self._flat.device == first.device)
        if ok:
            for spec, lin in self._named_linears():
                if lin.weight.data_ptr() != self._flat.data_ptr() + 4 * spec.w_off or \
                        lin.bias.data_ptr() != self._flat.data_ptr() + 4 * spec.b_off:
                    ok = False
                    break
        if not ok:
            if self._flat is not None and (self._flat.requires_grad or self._flat.grad is not None):
                raise RuntimeError(
                    "MLP.flat_params(): the parameters no longer alias the flat blob that Config.hip_flat_grads made a leaf "
                    "(model.to() / a dtype cast / re-pointed .data after flat_parameter()).  An optimiser built on the old "
                    "blob would silently stop training: call MLP.release_flat_parameter(), move the model, then rebuild "
                    "the optimiser on flat_parameter().")
            flat = torch.empty(self.num_params, dtype=torch.float32, device=first.device)
            with torch.no_grad():
                for spec, lin in self._named_linears():
                    n = spec.out_dim * spec.in_dim
                    flat[spec.w_off:spec.w_off + n].copy_(lin.weight.detach().reshape(-1))
                    flat[spec.b_off:spec.b_off + spec.out_dim].copy_(lin.bias.detach())
                    lin.weight.data = flat[spec.w_off:spec.w_off + n].view(spec.out_dim, spec.in_dim)
                    lin.bias.data = flat[spec.b_off:spec.b_off + spec.out_dim]
            self._flat = flat
            self._packed_key = None
        return self._flat

    def flat_parameter(self) -> torch.Tensor:
        """The canonical blob as ONE differentiable leaf (Config.hip_flat_grads): Model.__call__ then routes the level
        backward's gradient to `flat_parameter().grad` (one tensor, one accumulation per backward) instead of to the 46
        nn.Parameters -- whose autograd accumulation is ~180 small kernels per step -- and an optimiser built on
        `[mlp.flat_parameter()]` updates all of them in place (they are views of this blob)."""
        flat = self.flat_params()
        if not flat.requires_grad:
            flat.requires_grad_(True)
        return flat

    def release_flat_parameter(self) -> None:
        """Leave the flat-gradient mode: the blob stops being an autograd leaf and its gradient is dropped, so that a later
        per-parameter training step (or allreduce_gradients) cannot pick up a stale flat gradient."""
        if self._flat is not None:
            self._flat.grad = None
            if self._flat.requires_grad:
                self._flat.requires_grad_(False)

    def load_flat_params(self, blob):
        """Copy a flat blob (numpy / tensor) into the parameters: this module's own blob (`num_params` elements,
        state_dict order) or, for a variant, also a CANONICAL blob -- its embedded elements are taken, the rest ignored."""
        flat = self.flat_params()
        blob = torch.as_tensor(blob, dtype=torch.float32).to(flat.device).reshape(-1)
        if blob.numel() == self.canon_size and self._embed_index_np is not None:
            blob = blob[self.embed_index()]
        with torch.no_grad():
            flat.copy_(blob)
        self._packed_key = None

    def embed_index(self):
        """Variant only: LongTensor, position of every element of this module's blob in the canonical blob."""
        flat = self.flat_params()
        if self._embed_index is None or self._embed_index.device != flat.device:
            self._embed_index = torch.as_tensor(self._embed_index_np, device=flat.device)
        return self._embed_index

    def canonical_blob(self) -> torch.Tensor:
        """The parameters as the canonical Ref-NeRF blob the kernels and the C ABI take (layout.PARAM_SPECS): the module's
        own blob, or for a variant its embedding (absent heads / columns / dead units are zeros; one index_copy)."""
        flat = self.flat_params()
        if self._embed_index_np is None:
            return flat
        if self._canon is None or self._canon.device != flat.device:
            self._canon = torch.zeros(self.canon_size, dtype=torch.float32, device=flat.device)
            if len(self._identity_fill_np):          # shallower trunks: identity layers behind the real ones (layout.identity_fill)
                self._canon[torch.as_tensor(self._identity_fill_np, device=flat.device)] = 1.0
        with torch.no_grad():
            self._canon.index_copy_(0, self.embed_index(), flat.detach())
        return self._canon

    @property
    def ipe_groups(self) -> int:
        """cfg.ipe_groups of the level kernels: 0 = the built-in octahedron / 1 basis, else groups of three directions"""
        return 0 if self.ipe_basis_dirs == 3 else self.ipe_basis_dirs // 3

    def kernel_basis(self):
        """general basis only: the [3 G, 3] direction rows on the parameters' device (refnerf_pack_weights_basis)"""
        if self.ipe_groups == 0:
            return None
        dev = self.spatial_net[0].weight.device
        if self._basis_dev is None or self._basis_dev.device != dev:
            self._basis_dev = torch.as_tensor(self._basis_np, device=dev).contiguous()
        return self._basis_dev

    @property
    def kernel_dir_enc(self) -> int:
        return _hip.DIRENC_IDE if self.use_directional_enc else _hip.DIRENC_POSENC

    @property
    def kernel_roughness_bias(self) -> float:
        """roughness_bias the level runs with: without a roughness head the raw value is 0 and this bias drives the
        softplus to exactly 0 (internal/models.py:636: roughness = 0)."""
        return float(self.roughness_bias) if self.enable_pred_roughness else layout.ROUGHNESS_OFF_BIAS

    def ordered_parameters(self):
        """[w0, b0, w1, b1, ...] in canonical (state_dict) order."""
        out = []
        for _, lin in self._named_linears():
            out += [lin.weight, lin.bias]
        return out

    def _param_version(self):
        return tuple(p._version for p in self.parameters())

    def mark_updated(self) -> None:
        """Tell the weight-image cache that an optimiser step has been applied (call it after `optimizer.step()` when the
        optimiser is fused, i.e. does not bump the tensors' version counters): the next inference call re-packs once and
        later ones reuse that image.  Without it every inference call after a training forward re-packs (correct, 16 us)."""
        self._step_pending = False
        if self._packed:
            self._packed = {k: (buf, None) for k, (buf, _) in self._packed.items()}

    def packed_weights(self, precision: int, force: bool = False) -> torch.Tensor:
        """MFMA operand image of the current parameters for a precision mode (one cached buffer per mode, re-packed in
        place when the parameters changed).  Change detection = the tensors' version counters, plus: `force` (a training
        forward: an optimiser step follows it, and fused optimisers do not bump version counters) -- see mark_updated()."""
        flat = self.flat_params()
        key = (precision, flat.data_ptr(), self._param_version(), flat._version)
        if self._packed is None:
            self._packed = {}
        if force:
            # a training forward: an optimiser step follows at an unknown moment, and a fused optimiser leaves no trace in
            # the version counters.  Until the version counters move or mark_updated() says the step is done, every
            # image is packed "stale": inference calls in between (validation hooks, the noisy-ray pass) re-pack each
            # time (16 us) instead of trusting an image that may predate the step.
            self._step_pending = True
        elif getattr(self, "_step_pending", False) and self._packed_key is not None and key[2:] != self._packed_key[2:]:
            self._step_pending = False                      # the parameters' version counters moved: ordinary change detection works again
        pending = getattr(self, "_step_pending", False)
        buf, have = self._packed.get(precision, (None, None))
        if force or pending or have != key:
            buf = _hip.pack_weights(self.canonical_blob(), buf, precision, basis=self.kernel_basis())
            self._packed[precision] = (buf, None if (force or pending) else key)
        self._packed_key = key
        return buf

    def packed_specular_weights(self, precision: int) -> torch.Tensor:
        """`enable_pred_specular_density` (models.py:502-503,583-584,624-625): the extra head is a second Linear(net_width, 1) on
        the trunk's output with the density head's bias and activation, and nothing downstream reads it (the reference computes
        `specular_weights` from it and drops them, models.py:250-258).  The fused kernels have no 140th head row; the value comes
        from a SECOND launch on a weight image whose density head IS the specular-density head -- same trunk, same samples (the
        level's samples depend on the incoming step function only), its `density` output = softplus(raw_specular + density_bias).
        Re-packed on every call (the parameters may have moved; 20-60 us), into one cached buffer per image kind."""
        if not self.enable_pred_specular_density:
            raise ValueError("enable_pred_specular_density is off")
        blob = self.canonical_blob().detach().clone()
        spec = next(sp for sp in self.specs if sp.name == "raw_density")
        wpos = torch.arange(spec.w_off, spec.w_off + spec.out_dim * spec.in_dim, device=blob.device)
        bpos = torch.arange(spec.b_off, spec.b_off + spec.out_dim, device=blob.device)
        if self._embed_index_np is not None:
            idx = self.embed_index()
            wpos, bpos = idx[wpos], idx[bpos]
        with torch.no_grad():
            blob[wpos] = self.raw_specular_density.weight.detach().reshape(-1).to(blob.dtype)
            blob[bpos] = self.raw_specular_density.bias.detach().reshape(-1).to(blob.dtype)
        if getattr(self, "_packed_spec", None) is None:
            self._packed_spec = {}
        buf = _hip.pack_weights(blob, self._packed_spec.get(precision), precision, basis=self.kernel_basis())
        self._packed_spec[precision] = buf
        return buf

    def __call__(self, gaussians, viewdirs=None, imageplane=None):
        """Evaluate the MLP on caller-supplied Gaussians (models.py:533-750).

        gaussians = (means [..., n, 3], covs [..., n, 3, 3] or [..., n, 3]), viewdirs [..., 3].
        Returns the reference's `ray_results` dict: density [..., n]; rgb, normals (training
        mode, else None), normals_pred, grad_pred, tint, diffuse, specular [..., n, 3];
        roughness [..., n, 1].  One launch of the MLP stage kernel (refnerf_mlp_forward, the
        fused level kernel without resampling / compositing; f32 arithmetic).  This entry is
        forward-only: gradients flow through Model.__call__ (refnerf_level_backward)."""
        del imageplane                                   # unused by the reference as well (models.py:536)
        _hip.require_device()
        means, covs = gaussians
        if viewdirs is None:
            raise ValueError("the fused Ref-NeRF MLP needs viewdirs (use_viewdirs / use_reflections)")
        dev = self.spatial_net[0].weight.device
        means = torch.as_tensor(means, dtype=torch.float32, device=dev)
        covs = torch.as_tensor(covs, dtype=torch.float32, device=dev)
        batch = tuple(means.shape[:-2])
        n = means.shape[-2]
        full = covs.dim() == means.dim() + 1
        m = means.reshape(-1, n, 3)
        c = covs.reshape(-1, n, 3, 3) if full else covs.reshape(-1, n, 3)
        v = torch.as_tensor(viewdirs, dtype=torch.float32, device=dev).reshape(-1, 3)
        if v.shape[0] != m.shape[0]:
            raise ValueError("viewdirs must have one direction per ray of the Gaussians batch")
        cfg = _hip.default_cfg(
            n_samples=int(n), n_in=1, training=int(self.training), compute_extras=0,
            srgb_mapping=int(self.srgb_mapping), srgb_mapping_normalization=int(self.srgb_mapping_normalization),
            precision=_PREC["f32"], dir_enc=self.kernel_dir_enc, ipe_groups=self.ipe_groups, density_bias=float(self.density_bias), roughness_bias=self.kernel_roughness_bias,
            rgb_premultiplier=float(self.rgb_premultiplier), rgb_bias=float(self.rgb_bias),
            rgb_padding=float(self.rgb_padding))
        res = _hip.mlp_forward(self.packed_weights(cfg.precision), cfg, m, c, v)

        def rs(x, *tail):
            return x.reshape(batch + (n,) + tuple(tail))
        ray_results = dict(density=rs(res["density"]), rgb=rs(res["rgb"], 3))
        if not self.disable_density_normals:                   # models.py:735-748: keys follow the flags
            ray_results["normals"] = rs(res["normals"], 3) if self.training else None
        ray_results["normals_pred"] = rs(res["normals_pred"], 3)
        ray_results["grad_pred"] = rs(res["grad_pred"], 3)
        if self.use_specular_tint:
            ray_results["tint"] = rs(res["tint"], 3)
        ray_results["diffuse"] = rs(res["diffuse"], 3)
        ray_results["specular"] = rs(res["specular"], 3)
        if self.enable_pred_specular_density:                  # models.py:745-746 (key order as in the reference)
            res2 = _hip.mlp_forward(self.packed_specular_weights(cfg.precision), cfg, m, c, v)
            ray_results["specular_density"] = rs(res2["density"])
        if self.enable_pred_roughness:
            ray_results["roughness"] = rs(res["roughness"], 1)
        return ray_results


# Differentiable outputs of one training level.  The first five are consumed by the kernel directly;
# the per-ray composites of the second group are linear in (weights, history) (render.py:161-165,227-231)
# and are folded into per-sample seeds by _fold_ray_seeds; the third group are the per-sample history
# entries (models.py:731-750).  Everything else (sdist, bin_idx, the detached density normals, grad_pred,
# distance_mean / percentiles) is marked non-differentiable: no loss of the reference's train_utils
# differentiates them.
_KERNEL_KEYS = ("r_rgb", "weights", "normals_pred", "r_acc", "r_distance")
_RAY_KEYS = ("r_diffuse", "r_specular", "r_normals", "r_normals_pred", "r_tint", "r_roughness")
_SAMPLE_KEYS = ("density", "rgb", "diffuse", "specular", "tint", "roughness")
_DIFF_KEYS = _KERNEL_KEYS + _RAY_KEYS + _SAMPLE_KEYS


def _srgb_grad(x):
    """d/dx image.linear_to_srgb (image.py:51-59)."""
    eps = torch.finfo(torch.float32).eps
    hi = (211.0 / 200.0) * (5.0 / 12.0) * torch.clamp(x, min=eps) ** (-7.0 / 12.0)
    hi = torch.where(x > eps, hi, torch.zeros_like(x))
    return torch.where(x <= 0.0031308, torch.full_like(x, 323.0 / 25.0), hi)


def _linear_to_srgb(x):
    eps = torch.finfo(torch.float32).eps
    return torch.where(x <= 0.0031308, (323.0 / 25.0) * x, (211.0 * torch.clamp(x, min=eps) ** (5.0 / 12.0) - 11.0) / 200.0)


def _fold_ray_seeds(cfg, saved, g, g_weights, g_npred):
    """Per-ray seeds on the secondary composites -> per-sample seeds (the transpose of
    render.py:161-165 [diffuse / specular incl. the render-time map of :166-216, which for these two
    is a clip or clip(srgb(.)) without normalisation] and :227-231 [extras = sum_i w_i x_i]).
    Returns (g_weights, g_normals_pred, sample_seeds)."""
    w = saved["weights"]
    seeds = {k: g.get(k) for k in _SAMPLE_KEYS}

    def add(cur, x):
        return x if cur is None else cur + x

    mode = cfg.render_srgb_mode
    for rk, sk in (("r_diffuse", "diffuse"), ("r_specular", "specular")):
        gr = g.get(rk)
        if gr is None:
            continue
        x = saved[sk]
        acc = w.sum(dim=-1)
        bg_w = torch.clamp(1.0 - acc, min=0.0)
        pre = (w[..., None] * x).sum(dim=-2) + bg_w[..., None] * cfg.bg_rgb
        if mode in (_hip.SRGB_MODES["linear"], _hip.SRGB_MODES["norm_linear"]):
            gr = gr * ((pre >= 0.0) & (pre <= 1.0)).to(gr.dtype)
        elif mode in (_hip.SRGB_MODES["srgb"], _hip.SRGB_MODES["norm_srgb"]):
            y = _linear_to_srgb(pre)
            gr = gr * ((y >= 0.0) & (y <= 1.0)).to(gr.dtype) * _srgb_grad(pre)
        gw = (x * gr[:, None, :]).sum(dim=-1) - ((acc < 1.0).to(gr.dtype) * gr.sum(dim=-1) * cfg.bg_rgb)[:, None]
        g_weights = add(g_weights, gw)
        seeds[sk] = add(seeds[sk], w[..., None] * gr[:, None, :])
    for rk, sk in (("r_normals", "normals"), ("r_normals_pred", "normals_pred"), ("r_tint", "tint"),
                   ("r_roughness", "roughness")):
        gr = g.get(rk)
        if gr is None:
            continue
        x = saved[sk]
        if sk == "roughness":
            g_weights = add(g_weights, x * gr[:, None])
            seeds[sk] = add(seeds[sk], w * gr[:, None])
            continue
        g_weights = add(g_weights, (x * gr[:, None, :]).sum(dim=-1))
        if sk == "normals_pred":
            g_npred = add(g_npred, w[..., None] * gr[:, None, :])
        elif sk == "tint":
            seeds[sk] = add(seeds[sk], w[..., None] * gr[:, None, :])
        # the density normals are detached (models.py:603-609): r_normals only reaches the weights
    return g_weights, g_npred, {k: v for k, v in seeds.items() if v is not None}


class _LevelFunction(torch.autograd.Function):
    """One level of Model.__call__ in training mode as a single autograd node:
    forward = refnerf_level_forward(training=1), backward = refnerf_level_backward
    (what autograd replays through models.py:162-306 in the reference)."""

    @staticmethod
    def forward(ctx, mlp, cfg, rays, holder, sdist_in, weights_in, *params):
        # training kernels read the f32 image (it carries the bf16 chain ops); a training forward re-packs
        # unconditionally -- an optimiser step always changes the weights, but not every optimiser bumps the tensors'
        # version counters (fused Adam does not), and the pack is 16 us
        # (the split-f16 chains on the built-in basis: their own image -- forward + transposed operands as one chunk stream)
        image = _hip.level_image(cfg.precision, True, mlp.ipe_groups)        # the library's own rule (refnerf_level_image)
        # (the levels of ONE Model.__call__ share the image of their MLP: no optimiser step can fall between them)
        reuse = holder.get("call_images")
        packed = reuse.get((id(mlp), image)) if reuse is not None else None
        if packed is None:
            packed = mlp.packed_weights(image, force=True)
            if reuse is not None:
                reuse[(id(mlp), image)] = packed
        res = _hip.level_forward(packed, cfg, rays, sdist_in, weights_in, history=True, save_activations=True)
        ctx.mlp, ctx.cfg, ctx.rays, ctx.packed = mlp, cfg, rays, packed
        ctx.packed_key = mlp._packed_key
        ctx.set_materialize_grads(False)           # outputs without an upstream gradient arrive as None, not as zero tensors
        ctx.bwd_precision = holder.get("bwd_precision", _hip.PREC_F32)
        ctx.flat_mode = bool(holder.get("flat_mode", False))
        ctx.saved = {k: res.pop(k) if k in ("activations", "activations_format") else res[k]
                     for k in ("sdist", "density", "rgb", "weights", "activations", "activations_format", "diffuse",
                               "specular", "tint", "roughness", "normals", "normals_pred")}
        diff = tuple(k for k in _DIFF_KEYS if k in res)
        keys = diff + tuple(k for k in res if k not in diff)
        holder["keys"] = keys                      # autograd Functions return tuples: tell the caller the names
        ctx.diff_keys = diff
        outs = tuple(res[k] for k in keys)
        ctx.mark_non_differentiable(*outs[len(diff):])
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        mlp = ctx.mlp
        if ctx.saved is None:
            raise RuntimeError("this level's saved activations were released by its first backward "
                               "(a second backward through the same graph / retain_graph is not supported)")
        # (only the parameter-version part of the key: an inference call in ANOTHER precision between this level's forward
        # and backward -- a validation hook, the noisy-ray pass -- changes key[0] but not the weights the backward reads)
        if mlp._packed_key is None or mlp._packed_key[1:] != ctx.packed_key[1:]:
            raise _hip.HipLibraryError("parameters changed between the training forward and backward of a level")
        g = {k: v for k, v in zip(ctx.diff_keys, gouts) if v is not None}
        grads = torch.zeros(mlp.canon_size, dtype=torch.float32, device=ctx.saved["sdist"].device)
        g_rgb = g.get("r_rgb")
        if g_rgb is None:
            g_rgb = torch.zeros_like(ctx.saved["sdist"][:, :3])
        g_weights, g_npred, seeds = _fold_ray_seeds(ctx.cfg, ctx.saved, g, g.get("weights"), g.get("normals_pred"))
        flat_mode = ctx.flat_mode
        cfg = ctx.cfg
        if ctx.bwd_precision != cfg.precision:
            cfg = type(cfg).from_buffer_copy(cfg)      # same level, bf16 chains in the backward kernel
            cfg.precision = ctx.bwd_precision
        _hip.level_backward(ctx.packed, cfg, ctx.rays, ctx.saved, g_rgb, g_weights, g_npred, grads,
                            g_r_acc=g.get("r_acc"), g_r_distance=g.get("r_distance"), sample_seeds=seeds)
        ctx.saved = None                           # release the 17.6 KB/sample activation buffer
        if mlp._embed_index_np is not None:        # a variant: its parameters' elements of the canonical gradient
            grads = grads.index_select(0, mlp.embed_index())
        if flat_mode:                              # Config.hip_flat_grads: the blob itself is the differentiable input
            return (None, None, None, None, None, None, grads)
        out = []
        for spec in mlp.specs:                     # same order as MLP.ordered_parameters()
            n = spec.out_dim * spec.in_dim
            out.append(grads[spec.w_off:spec.w_off + n].view(spec.out_dim, spec.in_dim))
            out.append(grads[spec.b_off:spec.b_off + spec.out_dim])
        return (None, None, None, None, None, None) + tuple(out)


@configs.configurable
class NerfMLP(MLP):
    pass


@configs.configurable
class PropMLP(MLP):
    pass


_PREC = {"f32": _hip.PREC_F32, "bf16": _hip.PREC_BF16, "f16": _hip.PREC_F16, "f16x2": _hip.PREC_F16X2}
_TRAIN_FWD_PREC = ("f32", "f16x2", "bf16")   # MLP chains of the training forward: exact fp32 | split f16 (parity-grade, fp32 ACT rows) | bf16


class _Lean(threading.local):
    depth = 0


_LEAN = _Lean()


@contextlib.contextmanager
def lean_ray_history():
    """Inside this context an inference-mode Model.__call__ does not materialise the per-sample
    `ray_history` tensors its caller is going to drop (render_image keeps the renderings only): the
    kernel gets NULL pointers for them (84 B/sample of stores less), the dict entries are None; `sdist`,
    `weights` and what the vis rays need stay."""
    _LEAN.depth += 1
    try:
        yield
    finally:
        _LEAN.depth -= 1


@configs.configurable
class Model(nn.Module):
    """A mip-NeRF-360 style model holding the MLPs (models.py:50-321)."""

    def __init__(
            self,
            config: Any = None,
            num_prop_samples: int = 64,
            num_nerf_samples: int = 32,
            num_levels: int = 3,
            bg_intensity_range: Tuple[float] = (1., 1.),
            anneal_slope: float = 10,
            use_viewdirs: bool = True,
            raydist_fn: Callable[..., Any] = None,
            ray_shape: str = 'cone',
            disable_integration: bool = False,
            single_jitter: bool = True,
            dilation_bias: float = 0.0025,
            dilation_multiplier: float = 0.5,
            single_mlp: bool = False,
            resample_padding: float = 0.0,
            opaque_background: bool = False,
            init_s_near: float = 0.,
            init_s_far: float = 1.,
    ):
        super().__init__()
        for k, v in list(locals().items()):
            if k not in ("self", "__class__"):
                setattr(self, k, v)
        # models.py:120-123
        self.nerf_mlp = NerfMLP()
        self.prop_mlp = self.nerf_mlp if self.single_mlp else PropMLP()
        unsupported = {}
        self._raydist_code = self._raydist_enum(self.raydist_fn)      # raises for functions coord.construct_ray_warps rejects too
        if not self.use_viewdirs:
            unsupported["use_viewdirs"] = False
        if unsupported:
            raise ValueError(f"Model options outside the fused Ref-NeRF path: {unsupported}")

    @staticmethod
    def _raydist_enum(fn):
        """Model.raydist_fn -> REFNERF_RAYDIST_*: None, 'piecewise', or one of the torch functions coord.construct_ray_warps
        knows the inverse of (coord.py:84-92: reciprocal, log, exp, sqrt, square) -- as the callable itself or as the gin
        reference text ('@torch.reciprocal')."""
        if fn is None:
            return _hip.RAYDIST[None]
        name = fn if isinstance(fn, str) else getattr(fn, "__name__", None)
        if isinstance(name, str):
            name = name.lstrip("@").split(".")[-1].strip("()")
        if name not in _hip.RAYDIST:
            raise KeyError(name)                      # what inv_mapping[fn.__name__] raises in the reference
        return _hip.RAYDIST[name]

    @property
    def device(self):
        return next(self.parameters()).device

    def _specular_density(self, mlp: MLP, cfg, r, sdist_in, weights_in):
        """ray_results['specular_density'] of one level (MLP.packed_specular_weights): the level's inference kernel in
        Config.hip_precision on the image whose density head is the specular-density head, same incoming step function -> same
        samples; its `density` output.  Detached: no loss of the reference reads it (the head's gradient is None there too)."""
        import ctypes
        cfg2 = type(cfg)()
        ctypes.memmove(ctypes.byref(cfg2), ctypes.byref(cfg), ctypes.sizeof(cfg))
        cfg2.training, cfg2.compute_extras, cfg2.wgrad_mode = 0, 0, _hip.WGRAD_BF16X3
        cfg2.precision = _PREC[getattr(self.config, "hip_precision", "f32")]
        image = _hip.level_image(cfg2.precision, False, mlp.ipe_groups)
        with torch.no_grad():
            res = _hip.level_forward(mlp.packed_specular_weights(image), cfg2, r, sdist_in.detach(), weights_in.detach(), history=("density",))
        return res["density"]

    def _level_cfg(self, mlp: MLP, n_samples, n_in, train_frac, compute_extras):
        cfg = self.config
        if self.ray_shape not in ('cone', 'cylinder'):
            raise ValueError('ray_shape must be \'cone\' or \'cylinder\'')      # render.py:126
        if cfg.render_with_specular_density and not mlp.enable_pred_specular_density:
            raise ValueError('Specular density prediction from mlps should be enabled.')  # models.py:250-252
        # (with the head enabled the reference computes `specular_weights` here and never reads them, models.py:253-258:
        #  nothing to compute)
        if self.anneal_slope > 0:                                               # models.py:190-195
            s = self.anneal_slope
            anneal = (s * train_frac) / ((s - 1) * train_frac + 1)
        else:
            anneal = 1.
        if self.bg_intensity_range[0] == self.bg_intensity_range[1]:            # models.py:261-267
            bg = self.bg_intensity_range[0]
        else:
            bg = (self.bg_intensity_range[0] + self.bg_intensity_range[1]) / 2
        mode = cfg.srgb_mapping_type if cfg.srgb_mapping_when_rendering else 'none'
        if mode not in _hip.SRGB_MODES:
            raise ValueError('Mapping types are none, linear, norm_linear, srgb, norm_srgb')  # render.py:218
        prec = getattr(cfg, "hip_precision", "f32")
        if prec not in _PREC:
            raise ValueError("Config.hip_precision must be one of 'f32', 'f16x2', 'f16', 'bf16'")
        train_prec = getattr(cfg, "hip_train_precision", "f32")
        if train_prec not in _TRAIN_FWD_PREC:      # 'f16' is an inference mode of the level kernel
            raise ValueError("Config.hip_train_precision must be 'f32', 'f16x2' or 'bf16'")
        if mlp.ipe_groups and (train_prec if self.training else prec) not in ("f32", "f16x2"):
            raise ValueError(f"IPE basis '{mlp.basis_shape}' / {mlp.basis_subdivisions} ({mlp.ipe_basis_dirs} directions): the fused kernels run a "
                             "general basis in the parity-grade modes (Config.hip_precision / hip_train_precision / hip_bwd_precision = 'f32' or "
                             "'f16x2'); the plain bf16 / f16 throughput modes are built for 'octahedron' / 1")
        wgrad = {"f32": _hip.WGRAD_F32, "bf16x3": _hip.WGRAD_BF16X3, "f16": _hip.WGRAD_F16}.get(getattr(cfg, "hip_wgrad_mode", "bf16x3"))
        if wgrad is None:
            raise ValueError("Config.hip_wgrad_mode must be 'bf16x3', 'f16' or 'f32'")
        if wgrad == _hip.WGRAD_F16 and (not self.training or train_prec != "f16x2" or mlp.ipe_groups or _hip.LEGACY_F16X2_TRAIN):
            # 'f16' = the weight-gradient GEMM of the split-f16 training kernels with every operand at ONE half: inference levels and
            # the other chain modes (f32 / bf16 chains, a general IPE basis) run the GEMM that goes with them
            wgrad = _hip.WGRAD_BF16X3
        return _hip.default_cfg(
            n_samples=int(n_samples), n_in=int(n_in), training=int(self.training),
            compute_extras=int(bool(compute_extras)), srgb_mapping=int(mlp.srgb_mapping),
            srgb_mapping_normalization=int(mlp.srgb_mapping_normalization), render_srgb_mode=mode,
            opaque_background=int(self.opaque_background), ray_shape=0 if self.ray_shape == 'cone' else 1,
            precision=_PREC[train_prec] if self.training else _PREC[prec], wgrad_mode=wgrad, anneal=float(anneal), resample_padding=float(self.resample_padding),
            s_near=float(self.init_s_near), s_far=float(self.init_s_far), density_bias=float(mlp.density_bias),
            dir_enc=mlp.kernel_dir_enc, raydist=self._raydist_enum(self.raydist_fn), disable_integration=int(bool(self.disable_integration)),
            ipe_groups=mlp.ipe_groups,
            roughness_bias=mlp.kernel_roughness_bias, rgb_premultiplier=float(mlp.rgb_premultiplier),
            rgb_bias=float(mlp.rgb_bias), rgb_padding=float(mlp.rgb_padding), bg_rgb=float(bg))

    def __call__(self, rays, train_frac, compute_extras):
        """The Ref-NeRF model (models.py:129-321).

        Returns (renderings, ray_history): per level a dict of per-ray tensors
        and a dict of per-sample tensors, keys/shapes/dtypes as in the reference.
        """
        _hip.require_device()
        dev = self.device
        batch_shape = tuple(rays.origins.shape[:-1])

        def flat(x, c):
            return torch.as_tensor(x, dtype=torch.float32, device=dev).reshape(-1, c)
        r = {"origins": flat(rays.origins, 3), "directions": flat(rays.directions, 3),
             "viewdirs": flat(rays.viewdirs, 3), "radii": flat(rays.radii, 1).reshape(-1),
             "near": flat(rays.near, 1).reshape(-1), "far": flat(rays.far, 1).reshape(-1)}
        R = r["origins"].shape[0]
        # models.py:153-157
        sdist = torch.cat([torch.full((R, 1), float(self.init_s_near), device=dev),
                           torch.full((R, 1), float(self.init_s_far), device=dev)], dim=-1)
        weights = torch.ones((R, 1), device=dev)
        renderings, ray_history = [], []
        # diagnostic (not part of the reference's return value): the CDF bin index of every sample, per level --
        # north_star's "sample indices"; bench.py and the parity tests compare them between arithmetic modes
        self.last_bin_idx = []
        prod_num_samples = 1
        call_images = {}                       # weight images packed by this call's training levels: (MLP, image kind) -> buffer
        for i_level in range(self.num_levels):
            is_prop = i_level < (self.num_levels - 1)
            num_samples = self.num_prop_samples if is_prop else self.num_nerf_samples
            if num_samples <= 1:
                raise ValueError(f'num_samples must be > 1, is {num_samples}.')   # stepfun.py:234-235
            # models.py:167-186: after the first level optionally dilate the step function the next level
            # resamples from (a few torch ops on the detached [R, M] step function; the kernel then resamples
            # from 3M-2 intervals)
            dilation = self.dilation_bias + self.dilation_multiplier * (self.init_s_far - self.init_s_near) / prod_num_samples
            prod_num_samples *= num_samples
            if i_level > 0 and (self.dilation_bias > 0 or self.dilation_multiplier > 0):
                from . import stepfun
                with torch.no_grad():
                    sdist, weights = stepfun.max_dilate_weights(sdist.detach(), weights.detach(), dilation,
                                                                domain=(self.init_s_near, self.init_s_far), renormalize=True)
                    sdist, weights = sdist[..., 1:-1].contiguous(), weights[..., 1:-1].contiguous()
                if weights.shape[-1] > 512:
                    raise ValueError(f'dilated step function has {weights.shape[-1]} intervals; the fused resampler '
                                     'takes at most 512 (num_samples <= 171 per level with dilation)')
            mlp = self.prop_mlp if is_prop else self.nerf_mlp
            cfg = self._level_cfg(mlp, num_samples, weights.shape[-1], train_frac, compute_extras)
            sd_in, w_in = sdist, weights                     # this level's incoming step function (the specular-density launch re-reads it)
            if self.training and torch.is_grad_enabled():
                # one autograd node per level; sdist / resampling inputs are detached (models.py:205-216)
                mlp.flat_params()
                bwd_prec = getattr(self.config, "hip_bwd_precision", "f32")
                if bwd_prec not in _TRAIN_FWD_PREC:
                    raise ValueError("Config.hip_bwd_precision must be 'f32', 'f16x2' or 'bf16'")
                if mlp.ipe_groups and (bwd_prec not in ("f32", "f16x2") or cfg.wgrad_mode != _hip.WGRAD_BF16X3):
                    raise ValueError("a general IPE basis trains with Config.hip_bwd_precision = 'f32' or 'f16x2' and hip_wgrad_mode = 'bf16x3'")
                train_prec = getattr(self.config, "hip_train_precision", "f32")
                if bwd_prec == "f16x2" and train_prec == "bf16":
                    raise ValueError("Config.hip_bwd_precision = 'f16x2' reads split-f16 pair units or fp32 rows: use hip_train_precision 'f16x2' or 'f32'")
                if train_prec == "f16x2" and bwd_prec != "f16x2" and not mlp.ipe_groups:
                    # the split-f16 training forward saves its layer inputs as hi / lo pair units (REFNERF_ACT_F16X2), which the
                    # split-f16 backward and its f16 weight-gradient GEMM consume (a general IPE basis keeps fp32 rows)
                    raise ValueError("Config.hip_train_precision = 'f16x2' goes with hip_bwd_precision = 'f16x2' (its saved activations are "
                                     "split-f16 pair units)")
                if train_prec == "f16x2" and not mlp.ipe_groups and cfg.wgrad_mode == _hip.WGRAD_F32:
                    #  the split-f16 chains hand the weight-gradient GEMM 16-bit operands (ACT hi / lo pair units and
                    # one-half rows, DELTA one half per element + factors): there is no fp32-product GEMM on those
                    raise ValueError("Config.hip_wgrad_mode = 'f32' goes with the exact-fp32 chains (hip_train_precision = hip_bwd_precision = 'f32'); "
                                     "the 'f16x2' chains feed their own f16 weight-gradient GEMM (hip_wgrad_mode = 'bf16x3', the default, or 'f16')")
                flat_mode = bool(getattr(self.config, "hip_flat_grads", False))
                if not flat_mode and mlp._flat is not None and (mlp._flat.requires_grad or mlp._flat.grad is not None):
                    mlp.release_flat_parameter()            # flat mode was switched off: no stale .grad on the blob
                holder = {"bwd_precision": _PREC[bwd_prec], "flat_mode": flat_mode, "call_images": call_images}
                diff_inputs = (mlp.flat_parameter(),) if flat_mode else tuple(mlp.ordered_parameters())
                outs = _LevelFunction.apply(mlp, cfg, r, holder, sdist.detach(), weights.detach(), *diff_inputs)
                res = dict(zip(holder["keys"], outs))
            else:
                lean = _LEAN.depth > 0 and not self.training
                image = _hip.level_image(cfg.precision, bool(cfg.training), mlp.ipe_groups)
                # (a training-mode level without autograd -- torch.no_grad() around a model in train() mode: the noisy-ray pass, shard
                #  checks -- runs the SAME kernel as the differentiable one, bit for bit; the f16x2 training kernel keeps its sign words
                #  and bottleneck rows in the activation buffer, so it gets one and drops it)
                scratch_act = image == _hip.IMAGE_F16X2_TRAIN
                res = _hip.level_forward(mlp.packed_weights(image), cfg, r, sdist, weights,
                                         history=(("rgb",) if compute_extras else ()) if lean else True, save_activations=scratch_act)
                if scratch_act:
                    res.pop("activations", None)
                    res.pop("activations_format", None)
            sdist, weights = res["sdist"], res["weights"]
            self.last_bin_idx.append(res.get("bin_idx"))

            def rs(x, *tail):
                return x.reshape(batch_shape + tuple(tail))
            N = num_samples
            rendering = {"rgb": rs(res["r_rgb"], 3), "diffuse": rs(res["r_diffuse"], 3),
                         "specular": rs(res["r_specular"], 3), "distance": rs(res["r_distance"], 1),
                         "acc": rs(res["r_acc"])}
            if compute_extras:                                                   # render.py:227-254
                # extras = the ray_results keys that start with 'normals' or are 'roughness' / 'tint' (models.py:280-284)
                if self.training and not mlp.disable_density_normals:
                    rendering["normals"] = rs(res["r_normals"], 3)
                rendering["normals_pred"] = rs(res["r_normals_pred"], 3)
                if mlp.use_specular_tint:
                    rendering["tint"] = rs(res["r_tint"], 3)
                if mlp.enable_pred_roughness:
                    rendering["roughness"] = rs(res["r_roughness"], 1)
                rendering["distance_mean"] = rs(res["r_distance_mean"])
                pct = res["r_percentiles"]
                rendering["distance_percentile_5"] = rs(pct[:, 0].contiguous())
                rendering["distance_median"] = rs(pct[:, 1].contiguous())
                rendering["distance_percentile_95"] = rs(pct[:, 2].contiguous())
                n = self.config.vis_num_rays                                     # models.py:290-301
                rendering["ray_sdist"] = sdist[:n, :]
                rendering["ray_weights"] = weights[:n, :]
                rendering["ray_rgbs"] = res["rgb"][:n, :, :]
            renderings.append(rendering)
            def hist(k, *tail):
                return rs(res[k], *tail) if k in res else None      # None: training-only / lean_ray_history()
            ray_results = {"density": hist("density", N), "rgb": hist("rgb", N, 3),
                           "normals": hist("normals", N, 3) if self.training else None,
                           "normals_pred": hist("normals_pred", N, 3),
                           "grad_pred": hist("grad_pred", N, 3), "tint": hist("tint", N, 3),
                           "diffuse": hist("diffuse", N, 3), "specular": hist("specular", N, 3)}
            if mlp.enable_pred_specular_density:            # models.py:745-746: between 'specular' and 'roughness'
                ray_results["specular_density"] = None if (_LEAN.depth > 0 and not self.training) else \
                    rs(self._specular_density(mlp, cfg, r, sd_in, w_in), N)
            ray_results.update({"roughness": hist("roughness", N, 1),
                                "sdist": rs(sdist, N + 1), "weights": rs(weights, N)})      # (read-only for the next level: no copies)
            # the reference's dict only has the keys its flags produce (models.py:735-748)
            if mlp.disable_density_normals:
                del ray_results["normals"]
            if not mlp.use_specular_tint:
                del ray_results["tint"]
            if not mlp.enable_pred_roughness:
                del ray_results["roughness"]
            ray_history.append(ray_results)

        if compute_extras:                                                       # models.py:308-319
            ws = [x['ray_weights'] for x in renderings]
            rgbs = [x['ray_rgbs'] for x in renderings]
            final_rgb = torch.sum(rgbs[-1] * ws[-1][..., None], dim=-2)
            for i in range(len(rgbs) - 1):
                renderings[i]['ray_rgbs'] = torch.broadcast_to(final_rgb[:, None, :], rgbs[i].shape)
        return renderings, ray_history


def construct_model(rays, config):
    """models.py:324-340.  The reference needs one dummy forward to shape its
    LazyLinear layers; here the shapes are known at construction, so `rays` is
    only accepted for signature compatibility."""
    del rays
    return Model(config=config)


def render_image(render_fn: Callable[[utils.Rays], Tuple[List[Mapping[Text, torch.Tensor]], List[Any]]],
                 rays: utils.Rays, config: configs.Config, verbose: bool = True,
                 device=torch.device('cuda')) -> MutableMapping[Text, Any]:
    """Render all the pixels of an image in chunks (models.py:763-825)."""
    torch.cuda.synchronize()
    height, width = rays.origins.shape[:2]
    num_rays = height * width
    rays = rays.reshape(num_rays, -1)
    chunks = []
    with lean_ray_history():               # the per-sample ray_history of a chunk is dropped right below
        for idx0 in range(0, num_rays, config.render_chunk_size):
            chunk_rays = rays[idx0:idx0 + config.render_chunk_size]
            chunk_rays.to(device)
            chunk_renderings, _ = render_fn(chunk_rays)
            chunk_rendering = chunk_renderings[-1]
            for k in chunk_renderings[0]:
                if k.startswith('ray_'):
                    chunk_rendering[k] = [r[k] for r in chunk_renderings]
            chunks.append({k: utils.recursive_detach(v) for k, v in chunk_rendering.items()})
    rendering = utils.merge_chunks(chunks)
    for k, z in rendering.items():
        if not k.startswith('ray_'):
            rendering[k] = z.reshape((height, width) + z.shape[1:])
    keys = [k for k in rendering if k.startswith('ray_')]
    if keys:
        temp_num_rays = rendering[keys[0]][0].shape[0]
        ray_idx = torch.randperm(temp_num_rays)[:config.vis_num_rays]
        for k in keys:
            rendering[k] = [r[ray_idx.to(r.device)] for r in rendering[k]]
    return rendering
