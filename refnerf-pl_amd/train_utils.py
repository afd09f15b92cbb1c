"""Losses of the training step, host side.

Same names / signatures / arithmetic as the reference's internal/train_utils.py:
compute_data_loss (:33-88), compute_depth_smoothness_loss (:90-119), orientation_loss (:165-183),
interlevel_loss (:150-162), predicted_normal_loss (:186-204), noisy_consistency_loss (:207-279), noisy_distance_consistency_loss
(:282-310), accumulated_weights_loss (:313-316), weights_entropy_loss (:318-329) and the loss assembly of
NeRFSystem.training_step (nerf_system.py:77-188) as `training_losses`.  They are a few elementwise torch
ops on the level outputs; their gradients w.r.t. the renderings / ray_history entries are the seeds
refnerf_level_backward consumes (models.py::_LevelFunction).  Optimiser / LR-schedule plumbing is outside
the hot path.
"""
import collections

import torch


def srgb_to_linear(srgb, eps=None):
    """image.py:62-70."""
    if eps is None:
        eps = torch.finfo(torch.float32).eps
    linear0 = 25 / 323 * srgb
    linear1 = torch.clamp(((200 * srgb + 11) / 211), min=eps) ** (12 / 5)
    return torch.where(srgb <= 0.04045, linear0, linear1)


def _l2_normalize(x):
    """ref_utils.py:40-42"""
    eps = torch.finfo(torch.float32).eps
    return x / torch.sqrt(torch.clamp(torch.sum(x ** 2, dim=-1, keepdim=True), min=eps))


def _pixel_penalty(sq_err, config):
    """per-channel penalty of the data term: plain L2 or Charbonnier (train_utils.py:53-61)"""
    kind = config.data_loss_type
    if kind == 'mse':
        return sq_err
    if kind == 'charb':
        return torch.sqrt(sq_err + config.charb_padding ** 2)
    assert False


def _angular_error_deg(weight, a, b):
    """weighted mean angle between unit vectors, in degrees (ref_utils.py:45-50)"""
    lim = 1 - torch.finfo(torch.float32).eps
    angle = torch.arccos(torch.clip((a * b).sum(-1), -lim, lim))
    return (weight * angle).sum() / weight.sum() * 180.0 / torch.pi


def _coarse_fine(per_level, n_levels, coarse_mult, fine_mult):
    """sum over levels in level order with the coarse multiplier on all but the last level (the reference's accumulation order)"""
    acc = 0.
    for lvl, term in enumerate(per_level):
        acc = acc + (fine_mult if lvl >= n_levels - 1 else coarse_mult) * term
    return acc


def compute_data_loss(batch, renderings, rays, config):
    """Data loss terms for RGB, and the disparity / normal statistics (train_utils.py:33-88).  Returns (loss, stats)."""
    dev = renderings[0]['rgb'].device
    target = torch.as_tensor(batch.rgb, dtype=torch.float32, device=dev)[..., :3]
    if config.supervised_by_linear_rgb:
        target = srgb_to_linear(target)
    # per-ray weight of the residual (mosaic masks, multiscale up-weighting: rays.lossmult), one copy per channel
    ray_w = torch.broadcast_to(torch.as_tensor(rays.lossmult, dtype=torch.float32, device=dev), target.shape)
    if config.disable_multiscale_loss:
        ray_w = torch.ones_like(ray_w)
    norm = ray_w.sum()
    per_level, stats = [], collections.defaultdict(list)
    for out in renderings:
        sq_err = (out['rgb'] - target) ** 2
        stats['mses'].append((ray_w * sq_err).sum() / norm)
        per_level.append((ray_w * _pixel_penalty(sq_err, config)).sum() / norm)
        if config.compute_disp_metrics:                       # train_utils.py:62-67 (disparity from the mean distance)
            disparity = 1 / (1 + out['distance_mean'])
            stats['disparity_mses'].append(((disparity - torch.as_tensor(batch.disps, device=dev)) ** 2).mean())
        if config.compute_normal_metrics:                     # train_utils.py:69-84
            if 'normals' in out:
                w = out['acc'] * torch.as_tensor(batch.alphas, dtype=torch.float32, device=dev)
                stats['normal_maes'].append(_angular_error_deg(
                    w, _l2_normalize(out['normals']), _l2_normalize(torch.as_tensor(batch.normals, dtype=torch.float32, device=dev))))
            else:                                             # normals not computed (eval mode): NaN, as in the reference
                stats['normal_maes'].append(torch.tensor(float('nan'), device=dev))
    per_level = torch.stack(per_level)
    loss = config.data_coarse_loss_mult * torch.sum(per_level[:-1]) + config.data_loss_mult * per_level[-1]
    return loss, {k: torch.stack([x.detach() for x in v]) for k, v in stats.items()}


def orientation_loss(rays, model, ray_history, config):
    """Orientation regulariser of Ref-NeRF (train_utils.py:165-183): normals facing away from the camera, weighted by the
    sample's rendering weight."""
    terms = []
    for level in ray_history:
        normals = level[config.orientation_loss_target]
        if normals is None:
            raise ValueError('Normals cannot be None if orientation loss is on.')
        weights = level['weights']
        to_camera = -torch.as_tensor(rays.viewdirs, dtype=torch.float32, device=weights.device)
        facing = (normals * to_camera[..., None, :]).sum(dim=-1)
        terms.append(torch.mean((weights * torch.clamp(facing, max=0.0) ** 2).sum(dim=-1)))
    return _coarse_fine(terms, model.num_levels, config.orientation_coarse_loss_mult, config.orientation_loss_mult)


def predicted_normal_loss(model, ray_history, config):
    """Predicted-normal supervision of Ref-NeRF (train_utils.py:186-204): 1 - cos between the density normals and the predicted
    ones, weighted by the sample's rendering weight."""
    terms = []
    for level in ray_history:
        n_density, n_mlp = level['normals'], level['normals_pred']
        if n_density is None or n_mlp is None:
            raise ValueError('Predicted normals and gradient normals cannot be None if '
                             'predicted normal loss is on.')
        terms.append(torch.mean((level['weights'] * (1.0 - torch.sum(n_density * n_mlp, dim=-1))).sum(dim=-1)))
    return _coarse_fine(terms, model.num_levels, config.predicted_normal_coarse_loss_mult, config.predicted_normal_loss_mult)


class _FusedRefNerfLosses(torch.autograd.Function):
    """data (mse) + orientation + predicted-normal loss of ONE level as a single autograd node on the fused kernels
    refnerf_losses_forward / refnerf_losses_backward (Config.hip_fused_losses): what compute_data_loss,
    orientation_loss and predicted_normal_loss compute with ~25 elementwise ATen ops per level, in one pass each way."""

    @staticmethod
    def forward(ctx, rgb, weights, normals_pred, aux):
        from . import _hip
        rgb_c, w_c, np_c = rgb.detach().contiguous(), weights.detach().contiguous(), normals_pred.detach().contiguous()
        terms = _hip.losses_forward(rgb_c, aux["gt"], aux["lossmult"], w_c, aux["orient_n"], aux["normals"], np_c, aux["viewdirs"])
        sums = terms.sum(dim=0)
        ctx.set_materialize_grads(False)
        ctx.aux, ctx.t = aux, (rgb_c, w_c, np_c)
        per_term = sums * aux["scales"]
        ctx.mark_non_differentiable(sums)
        return per_term.sum(), per_term, sums

    @staticmethod
    def backward(ctx, g_total, g_terms, _g_sums):
        from . import _hip
        aux = ctx.aux
        rgb_c, w_c, np_c = ctx.t
        # upstream gradient of the three terms, on the device (no host sync): the total's plus each term's own
        if g_total is None and g_terms is None:
            return None, None, None, None
        if g_terms is None:
            up3 = g_total.to(torch.float32).expand(3)
        elif g_total is None:
            up3 = g_terms.to(torch.float32)
        else:
            up3 = g_terms.to(torch.float32) + g_total.to(torch.float32)
        g_rgb, g_w, g_np = _hip.losses_backward(rgb_c, aux["gt"], aux["lossmult"], w_c, aux["orient_n"], aux["orient_on_pred"],
                                                aux["normals"], np_c, aux["viewdirs"], 1.0, 1.0, 1.0,
                                                upstream=(up3 * aux["scales"]).contiguous())
        return g_rgb, g_w, g_np, None


def fused_losses_supported(config):
    """The fused path covers the loss set of the shipped refnerf configs: mse data term on the rendered rgb, orientation
    and predicted-normal regularisers."""
    return (config.data_loss_type == 'mse' and not config.supervised_by_linear_rgb and not config.compute_disp_metrics)


def fused_refnerf_losses(model, batch, rays, renderings, ray_history, config):
    """compute_data_loss + orientation_loss + predicted_normal_loss (train_utils.py:33-88,165-204) through the fused
    kernels: returns (data_loss, stats, orientation_loss, predicted_normal_loss) with the same values (fp32 summation
    order aside) and the same gradients into renderings['rgb'], ray_history['weights'], ray_history['normals_pred']."""
    dev = renderings[0]['rgb'].device
    f32 = dict(dtype=torch.float32, device=dev)
    gt = torch.as_tensor(batch.rgb, **f32)[..., :3].reshape(-1, 3).contiguous()
    lossmult = torch.as_tensor(rays.lossmult, **f32).reshape(-1).contiguous()
    if config.disable_multiscale_loss:
        lossmult = torch.ones_like(lossmult)
    denom = 3.0 * lossmult.sum()                      # sum of the lossmult broadcast to [R, 3]
    viewdirs = torch.as_tensor(rays.viewdirs, **f32).reshape(-1, 3).contiguous()
    R = gt.shape[0]
    inv_denom = 1.0 / denom                           # stays on the device: no host synchronisation in the step
    want_o = _any_positive(config, 'orientation_coarse_loss_mult', 'orientation_loss_mult')
    want_n = _any_positive(config, 'predicted_normal_coarse_loss_mult', 'predicted_normal_loss_mult')
    data, orient, normal, mses = [], [], [], []
    for i, (rendering, hist) in enumerate(zip(renderings, ray_history)):
        N = hist['weights'].shape[-1]
        fine = i == model.num_levels - 1
        orient_n = None
        if want_o:
            orient_n = hist[config.orientation_loss_target]
            if orient_n is None:
                raise ValueError('Normals cannot be None if orientation loss is on.')
            orient_n = orient_n.detach().reshape(R, N, 3).contiguous()
        normals = None
        if want_n:
            if hist['normals'] is None or hist['normals_pred'] is None:
                raise ValueError('Predicted normals and gradient normals cannot be None if predicted normal loss is on.')
            normals = hist['normals'].detach().reshape(R, N, 3).contiguous()
        scales = torch.stack([(config.data_loss_mult if fine else config.data_coarse_loss_mult) * inv_denom,
                              torch.full_like(inv_denom, (config.orientation_loss_mult if fine else config.orientation_coarse_loss_mult) / R),
                              torch.full_like(inv_denom, (config.predicted_normal_loss_mult if fine else config.predicted_normal_coarse_loss_mult) / R)])
        aux = dict(gt=gt, lossmult=lossmult, viewdirs=viewdirs, orient_n=orient_n, normals=normals, scales=scales,
                   orient_on_pred=config.orientation_loss_target == 'normals_pred')
        _, per_term, sums = _FusedRefNerfLosses.apply(rendering['rgb'].reshape(R, 3), hist['weights'].reshape(R, N),
                                                      hist['normals_pred'].reshape(R, N, 3), aux)
        data.append(per_term[0]); orient.append(per_term[1]); normal.append(per_term[2])
        mses.append(sums[0] * inv_denom)
    stats = {'mses': torch.stack([m.detach() for m in mses])}
    return (torch.stack(data).sum(), stats, torch.stack(orient).sum() if want_o else None,
            torch.stack(normal).sum() if want_n else None)


def interlevel_loss(ray_history, config):
    """Interlevel (proposal) loss of mip-NeRF 360 (train_utils.py:150-162): the proposal levels' weights must
    envelope the final level's; the final level is detached, so only the proposal MLP is trained by it."""
    from . import stepfun
    c = ray_history[-1]['sdist'].detach()
    w = ray_history[-1]['weights'].detach()
    total = 0.
    for ray_results in ray_history[:-1]:
        total = total + torch.mean(stepfun.lossfun_outer(c, w, ray_results['sdist'], ray_results['weights']))
    return config.interlevel_loss_mult * total


def _level_mult(i, model, coarse, fine):
    return coarse if i < model.num_levels - 1 else fine


def compute_depth_smoothness_loss(renderings, config):
    """Edge-aware depth smoothness over [..., H, W, .] patches (train_utils.py:90-119)."""
    def l1(x):
        return torch.mean(torch.abs(x))

    def bilateral(x):
        return torch.exp(-torch.abs(x).mean(-1, keepdim=True))
    per_level = []
    for rendering in renderings:
        depths = rendering['distance']
        acc00 = rendering['acc'].detach()[..., :-1, :-1, None]
        guide = rendering['rgb'].detach()
        v00, v01, v10 = depths[..., :-1, :-1, :], depths[..., :-1, 1:, :], depths[..., 1:, :-1, :]
        w01 = bilateral(guide[..., :-1, :-1, :] - guide[..., :-1, 1:, :])
        w10 = bilateral(guide[..., :-1, :-1, :] - guide[..., 1:, :-1, :])
        per_level.append((l1(acc00 * w01 * (v00 - v01) ** 2) + l1(acc00 * w10 * (v00 - v10) ** 2)) / 2)
    per_level = torch.stack(per_level)
    return config.depth_smoothness_coarse_loss_mult * torch.sum(per_level[:-1]) + \
        config.depth_smoothness_loss_mult * per_level[-1]


def _colour_consistency(kind, clean, noisy, mask):
    """One of the three colour-consistency measures of train_utils.py:224-250 between the clean ray
    [n,1,3] and its noisy copies [n,a,3]; `mask` [n,1] selects the rays that count."""
    if kind == 'mse':
        per_ray = ((clean - noisy) ** 2).mean(dim=1, keepdim=True)
    elif kind == 'avg_mse':
        per_ray = ((clean - noisy.mean(dim=1, keepdim=True)) ** 2).mean(dim=1, keepdim=True)
    elif kind == 'var':
        both = torch.cat([clean, noisy], dim=1)
        return both.var(dim=1, keepdim=True).mean(dim=-1, keepdim=True).sum(dim=-1)[mask].mean()
    else:
        raise ValueError(f'unknown consistency loss type {kind!r}')
    return per_ray.sum(dim=-1)[mask].mean()


def noisy_consistency_loss(model, renderings, renderings_noise, config, warmup_ratio=1.):
    """Diffuse / specular / normal consistency between each of the first n rays and its
    `sample_noise_angles` perturbed copies (train_utils.py:207-279).  Returns the three totals."""
    n = config.sample_noise_size // config.patch_size ** 2
    a = config.sample_noise_angles
    totals = [0., 0., 0.]
    for i, (clean, noisy) in enumerate(zip(renderings, renderings_noise)):
        def group(x):
            return x.reshape((n, a) + tuple(x.shape[1:]))
        mask = clean['acc'][:n, None] > config.acc_threshold_for_consistency_loss
        diffuse = _colour_consistency(config.consistency_diffuse_loss_type, clean['diffuse'][:n, None],
                                      group(noisy['diffuse']), mask)
        # the specular term is maximised: view-dependent colour is pushed into the specular branch
        specular = -_colour_consistency(config.consistency_specular_loss_type, clean['specular'][:n, None],
                                        group(noisy['specular']), mask)
        if clean.get('normals') is None or clean.get('normals_pred') is None:
            raise ValueError('Predicted normals and gradient normals cannot be None if consistency loss is on.')
        target = config.consistency_normal_loss_target
        if target not in ('normals', 'normals_pred'):
            raise ValueError('Given an unknown type of consistency_normal_loss_target.')
        cos = torch.sum(clean[target][:n, None] * group(noisy[target]), dim=-1)
        normal = (1.0 - cos).mean(dim=1, keepdim=True)[mask].mean()
        for j, (term, coarse, fine) in enumerate((
                (diffuse, config.consistency_diffuse_coarse_loss_mult, config.consistency_diffuse_loss_mult),
                (specular, config.consistency_specular_coarse_loss_mult, config.consistency_specular_loss_mult),
                (normal, config.consistency_normal_coarse_loss_mult, config.consistency_normal_loss_mult))):
            totals[j] = totals[j] + warmup_ratio * _level_mult(i, model, coarse, fine) * term
    return tuple(totals)


def noisy_distance_consistency_loss(model, rays, noisy_rays, renderings, renderings_noise, config, warmup_ratio=1.):
    """The clean ray and its noisy copies should hit the same 3-D point (train_utils.py:282-310)."""
    n = config.sample_noise_size // config.patch_size ** 2
    a = config.sample_noise_angles
    total = 0.
    for i, (clean, noisy) in enumerate(zip(renderings, renderings_noise)):
        dev = clean['distance'].device

        def f32(x):
            return torch.as_tensor(x, dtype=torch.float32, device=dev)
        hit = f32(rays.origins)[:n, None] + f32(rays.directions)[:n, None] * clean['distance'][:n, None]
        hit_n = f32(noisy_rays.origins).reshape(n, a, 3) + \
            f32(noisy_rays.directions).reshape(n, a, 3) * noisy['distance'].reshape(n, a, 1)
        mask = clean['acc'][:n, None] > config.acc_threshold_for_consistency_loss
        if config.consistency_distance_loss_type != 'mse':
            raise ValueError('consistency_distance_loss_type must be mse')
        loss = ((hit - hit_n) ** 2).mean(dim=1, keepdim=True).sum(dim=-1)[mask].mean()
        total = total + warmup_ratio * _level_mult(i, model, config.consistency_distance_coarse_loss_mult,
                                                   config.consistency_distance_loss_mult) * loss
    return total


def accumulated_weights_loss(renderings, config):
    """Pushes the fine level's accumulated opacity towards 1 (train_utils.py:313-316)."""
    return config.accumulated_weights_loss_mult * ((1 - renderings[-1]['acc']) ** 2).mean()


def weights_entropy_loss(model, renderings, ray_history, config, warmup_ratio):
    """Entropy of the compositing weights on rays that hit something (train_utils.py:318-329)."""
    total = 0.
    for i, (rendering, ray_results) in enumerate(zip(renderings, ray_history)):
        w = ray_results['weights'][rendering['acc'] > config.acc_threshold_for_weights_entropy_loss]
        loss = (-w * (w + 1e-10).log()).sum(dim=-1).mean()
        total = total + warmup_ratio * _level_mult(i, model, config.weights_entropy_coarse_loss_mult,
                                                   config.weights_entropy_loss_mult) * loss
    return total


def consistency_warmup_ratio(config, global_step):
    """Warm-up / decay schedule of the consistency terms (nerf_system.py:92-108)."""
    if config.consistency_warmup_steps > config.consistency_decay_steps:
        raise ValueError("Consistency loss decay should be after whole warmup.")
    ratio = 1.
    if 0. < config.consistency_warmup_steps <= 1.:
        ratio = min(1., global_step / (config.consistency_warmup_steps * config.max_steps))
    if 0. < config.consistency_decay_steps <= 1. and global_step >= config.consistency_decay_steps * config.max_steps:
        left = config.max_steps - global_step
        ratio = max(0., left / (config.max_steps - config.consistency_decay_steps * config.max_steps))
    return ratio


def _any_positive(config, *names):
    return any(getattr(config, n) > 0 for n in names)


_CONSISTENCY_MULTS = tuple(f"consistency_{k}_{lvl}loss_mult" for k in ("diffuse", "specular", "normal")
                           for lvl in ("coarse_", ""))


def wants_noisy_pass(config):
    """nerf_system.py:110-116: the second, perturbed-ray forward is needed."""
    return config.sample_noise_size > 0 and _any_positive(config, *_CONSISTENCY_MULTS)


def training_losses(model, batch, rays, config, train_frac=1.0, global_step=None, noisy_rays=None):
    """Forward(s) + every loss term of NeRFSystem.training_step (nerf_system.py:77-188).

    Runs Model.__call__ on `rays` and, when a consistency term is on, on the perturbed rays of
    sample_utils.sample_noisy_rays (or on `noisy_rays` if given: reproducible tests).  Returns
    (total, dict of terms, stats, aux) with aux = dict(renderings, ray_history, noisy_rays, ...)."""
    from . import sample_utils
    compute_extras = bool(config.compute_disp_metrics or config.compute_normal_metrics or config.sample_noise_size > 0)
    renderings, ray_history = model(rays, train_frac, compute_extras)
    step = config.max_steps if global_step is None else global_step
    ratio = consistency_warmup_ratio(config, step)
    renderings_noise = None
    if wants_noisy_pass(config):
        if config.patch_size ** 2 > config.sample_noise_size:
            raise ValueError(f'Patch size {config.patch_size}^2 too large for '
                             f'sampling noise view points {config.sample_noise_size}')
        if noisy_rays is None:
            noisy_rays = sample_utils.sample_noisy_rays(
                rays, renderings[-1], config.sample_angle_range, config.sample_noise_size // config.patch_size ** 2,
                config.sample_noise_angles, ratio)
        renderings_noise, _ = model(noisy_rays, train_frac, True)
    total, losses, stats = compute_losses(model, batch, rays, renderings, ray_history, config,
                                          renderings_noise=renderings_noise, noisy_rays=noisy_rays, warmup_ratio=ratio)
    return total, losses, stats, dict(renderings=renderings, ray_history=ray_history, noisy_rays=noisy_rays,
                                      renderings_noise=renderings_noise, warmup_ratio=ratio)


def compute_losses(model, batch, rays, renderings, ray_history, config, renderings_noise=None, noisy_rays=None,
                   warmup_ratio=1.):
    """The loss assembly of NeRFSystem.training_step (nerf_system.py:118-180) on already computed
    renderings: returns (total, dict of terms, stats)."""
    losses = {}
    fused = getattr(config, 'hip_fused_losses', False) and fused_losses_supported(config) and renderings[0]['rgb'].is_cuda
    if fused:      # opt-in: the three Ref-NeRF terms of every level through refnerf_losses_forward / _backward
        data_loss, stats, o_loss, n_loss = fused_refnerf_losses(model, batch, rays, renderings, ray_history, config)
    else:
        data_loss, stats = compute_data_loss(batch, renderings, rays, config)
    losses['data'] = data_loss
    if config.interlevel_loss_mult > 0:
        losses['interlevel'] = interlevel_loss(ray_history, config)
    if _any_positive(config, 'orientation_coarse_loss_mult', 'orientation_loss_mult'):
        losses['orientation'] = o_loss if fused else orientation_loss(rays, model, ray_history, config)
    if _any_positive(config, 'predicted_normal_coarse_loss_mult', 'predicted_normal_loss_mult'):
        losses['predicted_normals'] = n_loss if fused else predicted_normal_loss(model, ray_history, config)
    if config.patch_size > 1 and _any_positive(config, 'depth_smoothness_coarse_loss_mult', 'depth_smoothness_loss_mult'):
        losses['smoothness'] = compute_depth_smoothness_loss(renderings, config)
    if wants_noisy_pass(config):
        if renderings_noise is None:
            raise ValueError('the consistency losses need the renderings of the noisy rays (training_losses)')
        (losses['diffuse_consistency'], losses['specular_consistency'],
         losses['normals_consistency']) = noisy_consistency_loss(model, renderings, renderings_noise, config, warmup_ratio)
    if config.accumulated_weights_loss_mult > 0:
        losses['acc'] = accumulated_weights_loss(renderings, config)
    if _any_positive(config, 'consistency_distance_loss_mult', 'consistency_distance_coarse_loss_mult'):
        if renderings_noise is None or noisy_rays is None:
            raise ValueError('the distance consistency loss needs the noisy rays and their renderings')
        losses['distance_consistency'] = noisy_distance_consistency_loss(
            model, rays, noisy_rays, renderings, renderings_noise, config, warmup_ratio)
    if _any_positive(config, 'weights_entropy_loss_mult', 'weights_entropy_coarse_loss_mult'):
        losses['weights_entropy'] = weights_entropy_loss(model, renderings, ray_history, config, warmup_ratio)
    total = torch.sum(torch.stack([torch.as_tensor(v, dtype=torch.float32, device=data_loss.device)
                                   for v in losses.values()]))
    if getattr(config, 'hip_check_finite', True):
        _FINITE_GUARD.watch(total, config)
    return total, losses, stats


class _FiniteGuard:
    """Loud failure for the operand-range limit of the 16-bit chain modes (include/refnerf_hip.h: beyond |x| = 65504 a split-f16
    unit becomes inf - inf and the level's outputs NaN).  No per-step synchronisation: every total is reduced to one flag on
    the device and copied to pinned host memory asynchronously; the flag of step k is read when step k + 1 asks (its copy has
    long completed by then), or on `flush()`.  Raises FloatingPointError naming the knobs that lift the limit."""

    def __init__(self):
        self.pending = None      # (host flag, event, step number)
        self.step = 0

    def _raise(self, step, config):
        raise FloatingPointError(
            f"non-finite training loss at step {step} of this process (hip_train_precision = {getattr(config, 'hip_train_precision', '?')!r}, "
            f"hip_bwd_precision = {getattr(config, 'hip_bwd_precision', '?')!r}): the split-f16 / f16 / bf16 chains hold operands up to 65504 only -- "
            "set Config.hip_precision = Config.hip_train_precision = Config.hip_bwd_precision = 'f32' (exact fp32 MFMA chains, no "
            "operand-range limit), or Config.hip_check_finite = False to train on regardless")

    def flush(self, config=None):
        if self.pending is not None:
            flag, event, step = self.pending
            self.pending = None
            if event is not None:
                event.synchronize()
            if not bool(flag.item()):
                self._raise(step, config)

    def watch(self, total, config):
        self.flush(config)                      # the previous step's flag: its copy was queued a whole step ago
        self.step += 1
        ok = torch.isfinite(total.detach())
        if total.is_cuda:
            flag = torch.empty((), dtype=torch.bool, pin_memory=True)
            flag.copy_(ok, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            self.pending = (flag, event, self.step)
        elif not bool(ok):
            self._raise(self.step, config)


_FINITE_GUARD = _FiniteGuard()


def flush_finite_check(config=None):
    """raise now if the LAST watched training loss was not finite (call at the end of a run / before a checkpoint)"""
    _FINITE_GUARD.flush(config)
