"""The three Ref-NeRF losses of the training step, host side.

Same names / signatures / arithmetic as the reference's
internal/train_utils.py:33-88 (compute_data_loss), :165-183 (orientation_loss)
and :186-204 (predicted_normal_loss).  They are a few elementwise torch ops on
the level outputs; their gradients w.r.t. renderings['rgb'], ray_history
['weights'] and ray_history['normals_pred'] are what refnerf_level_backward
consumes.  Everything else in the reference's train_utils (other regularisers,
optimiser / LR schedule plumbing) is outside the hot path.
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


def compute_data_loss(batch, renderings, rays, config):
    """Data loss terms for RGB (train_utils.py:33-88).  Returns (loss, stats)."""
    data_losses = []
    stats = collections.defaultdict(lambda: [])
    dev = renderings[0]['rgb'].device
    gt_all = torch.as_tensor(batch.rgb, dtype=torch.float32, device=dev)[..., :3]
    lossmult = torch.as_tensor(rays.lossmult, dtype=torch.float32, device=dev)
    lossmult = torch.broadcast_to(lossmult, gt_all.shape)
    if config.disable_multiscale_loss:
        lossmult = torch.ones_like(lossmult)
    for rendering in renderings:
        gt_rgb = gt_all
        if config.supervised_by_linear_rgb:
            gt_rgb = srgb_to_linear(gt_rgb)
        resid_sq = (rendering['rgb'] - gt_rgb) ** 2
        denom = lossmult.sum()
        stats['mses'].append((lossmult * resid_sq).sum() / denom)
        if config.data_loss_type == 'mse':
            data_loss = resid_sq
        elif config.data_loss_type == 'charb':
            data_loss = torch.sqrt(resid_sq + config.charb_padding ** 2)
        else:
            assert False
        data_losses.append((lossmult * data_loss).sum() / denom)
        if config.compute_disp_metrics:
            disp = 1 / (1 + rendering['distance_mean'])
            stats['disparity_mses'].append(((disp - torch.as_tensor(batch.disps, device=dev)) ** 2).mean())
    data_losses = torch.stack(data_losses)
    loss = config.data_coarse_loss_mult * torch.sum(data_losses[:-1]) + config.data_loss_mult * data_losses[-1]
    stats = {k: torch.stack([x.detach() for x in v]) for k, v in stats.items()}
    return loss, stats


def orientation_loss(rays, model, ray_history, config):
    """Orientation regulariser of Ref-NeRF (train_utils.py:165-183)."""
    total_loss = 0.
    for i, ray_results in enumerate(ray_history):
        w = ray_results['weights']
        n = ray_results[config.orientation_loss_target]
        if n is None:
            raise ValueError('Normals cannot be None if orientation loss is on.')
        v = -torch.as_tensor(rays.viewdirs, dtype=torch.float32, device=w.device)
        n_dot_v = (n * v[..., None, :]).sum(dim=-1)
        loss = torch.mean((w * torch.clamp(n_dot_v, max=0.0) ** 2).sum(dim=-1))
        if i < model.num_levels - 1:
            total_loss += config.orientation_coarse_loss_mult * loss
        else:
            total_loss += config.orientation_loss_mult * loss
    return total_loss


def predicted_normal_loss(model, ray_history, config):
    """Predicted-normal supervision of Ref-NeRF (train_utils.py:186-204)."""
    total_loss = 0.
    for i, ray_results in enumerate(ray_history):
        w = ray_results['weights']
        n = ray_results['normals']
        n_pred = ray_results['normals_pred']
        if n is None or n_pred is None:
            raise ValueError('Predicted normals and gradient normals cannot be None if '
                             'predicted normal loss is on.')
        loss = torch.mean((w * (1.0 - torch.sum(n * n_pred, dim=-1))).sum(dim=-1))
        if i < model.num_levels - 1:
            total_loss += config.predicted_normal_coarse_loss_mult * loss
        else:
            total_loss += config.predicted_normal_loss_mult * loss
    return total_loss


def compute_losses(model, batch, rays, renderings, ray_history, config):
    """The loss assembly of NeRFSystem.training_step restricted to the Ref-NeRF
    terms (nerf_system.py:95-140): returns (total, dict of terms)."""
    losses = {}
    data_loss, stats = compute_data_loss(batch, renderings, rays, config)
    losses['data'] = data_loss
    if config.orientation_coarse_loss_mult > 0 or config.orientation_loss_mult > 0:
        losses['orientation'] = orientation_loss(rays, model, ray_history, config)
    if config.predicted_normal_coarse_loss_mult > 0 or config.predicted_normal_loss_mult > 0:
        losses['predicted_normals'] = predicted_normal_loss(model, ray_history, config)
    return sum(losses.values()), losses, stats
