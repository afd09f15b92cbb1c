"""Perturbed ("noisy") rays for the consistency regularisers -- host mirror of the reference's
internal/sample_utils.py (euler_angles_to_matrix :4-38, sample_noisy_rays :40-79).

For each of the first `sample_noise_size` rays and each of `sample_noise_angles` random small
rotations T, the new ray looks at the same surface point p = o + distance * d from the rotated
direction: d' = T d, o' = p - distance * d'.  A few tiny torch ops per training step; the rays then go
through the fused HIP path like any other batch.
"""
import math

import torch

from . import utils


def euler_angles_to_matrix(euler_angles: torch.Tensor) -> torch.Tensor:
    """XYZ Euler angles [..., 3] (radians) -> rotation matrices [..., 3, 3] = Rx @ Ry @ Rz."""
    if euler_angles.dim() == 0 or euler_angles.shape[-1] != 3:
        raise ValueError("Invalid input euler angles.")
    ax, ay, az = torch.unbind(euler_angles, -1)
    one, zero = torch.ones_like(ax), torch.zeros_like(ax)

    def mat(*rows):
        return torch.stack(rows, -1).reshape(ax.shape + (3, 3))
    cx, sx, cy, sy, cz, sz = torch.cos(ax), torch.sin(ax), torch.cos(ay), torch.sin(ay), torch.cos(az), torch.sin(az)
    rx = mat(one, zero, zero, zero, cx, -sx, zero, sx, cx)
    ry = mat(cy, zero, sy, zero, one, zero, -sy, zero, cy)
    rz = mat(cz, -sz, zero, sz, cz, zero, zero, zero, one)
    return rx @ ry @ rz


@torch.no_grad()
def sample_noisy_rays(rays: utils.Rays, rendering: dict, sample_angle_range: float = 0.,
                      sample_noise_size: int = 128, sample_noise_angles: int = 1,
                      warmup_ratio: float = 1., rotations=None) -> utils.Rays:
    """sample_utils.py:40-79.  `rotations` ([angles,3,3]) overrides the random draw (tests)."""
    dev = rendering['distance'].device
    if rotations is None:
        hi = sample_angle_range / 180 * math.pi * warmup_ratio
        angles = torch.zeros(sample_noise_angles * 3, device=dev).uniform_(0, hi).reshape(-1, 3)
        rotations = euler_angles_to_matrix(angles)
    rotations = torch.as_tensor(rotations, dtype=torch.float32, device=dev)
    n = min(sample_noise_size, len(rendering['distance']))

    def f32(x):
        return torch.as_tensor(x, dtype=torch.float32, device=dev)

    def tiled(x):
        return torch.cat([f32(x)[:n]] * sample_noise_angles)
    distance = rendering['distance']
    if distance.dim() == f32(rays.origins).dim() - 1:
        distance = distance[..., None]
    elif distance.dim() != f32(rays.origins).dim():
        raise ValueError('The dimension of distance is wrong.')
    distance = torch.cat([distance[:n]] * sample_noise_angles)
    viewdirs_ = torch.cat([f32(rays.viewdirs)[:n] @ T.T for T in rotations])
    directions_ = torch.cat([f32(rays.directions)[:n] @ T.T for T in rotations])
    origins_ = tiled(rays.origins) + distance * tiled(rays.directions) - distance * directions_
    return utils.Rays(origins=origins_, directions=directions_, viewdirs=viewdirs_, radii=tiled(rays.radii),
                      imageplane=tiled(rays.imageplane), lossmult=tiled(rays.lossmult), near=tiled(rays.near),
                      far=tiled(rays.far), cam_idx=tiled(rays.cam_idx))
