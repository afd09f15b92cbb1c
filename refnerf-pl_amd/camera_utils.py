"""Ray generation in front of the hot path, on the device.

Mirrors the call surface of the reference's internal/camera_utils.py for the part
the Ref-NeRF configs use: `pixels_to_rays` (:502-614), `cast_ray_batch` (:617-670),
`cast_pinhole_rays` (:673-697), `pixel_coordinates` / `get_pixtocam` (:380-406).
Perspective cameras without lens distortion; the NDC conversion (:31-97) is
included.  The arithmetic runs in one HIP kernel (refnerf_pixels_to_rays): pixel
indices in, the `Rays` fields out, so whole-image rendering neither casts rays
with numpy on the host nor copies 64 B/ray over PCIe.
"""
import enum

import numpy as np
import torch

from . import _hip, utils


class ProjectionType(enum.Enum):
    """camera_utils.py:493-496"""
    PERSPECTIVE = 'perspective'
    FISHEYE = 'fisheye'


def intrinsic_matrix(fx, fy, cx, cy):
    """camera_utils.py:379-386"""
    return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.]])


def get_pixtocam(focal, width, height):
    """Inverse intrinsic matrix for a perfect pinhole camera (camera_utils.py:389-393)."""
    return np.linalg.inv(intrinsic_matrix(focal, focal, width * .5, height * .5)).astype(np.float32)


def pixel_coordinates(width, height, device):
    """Tuple of the x and y integer coordinates for a grid of pixels (camera_utils.py:396-400)."""
    y, x = torch.meshgrid(torch.arange(height, dtype=torch.int32, device=device),
                          torch.arange(width, dtype=torch.int32, device=device), indexing='ij')
    return x, y


def _dev(x, device, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=dtype).to(device)


def pixels_to_rays(pix_x_int, pix_y_int, pixtocams, camtoworlds, distortion_params=None, pixtocam_ndc=None,
                   camtype=ProjectionType.PERSPECTIVE, xnp=torch, device=None):
    """camera_utils.pixels_to_rays: returns (origins, directions, viewdirs [SH,3], radii [SH,1],
    imageplane [SH,2]) as device tensors.  `pixtocams` / `camtoworlds` are one camera ([3,3] / [3,4])
    or one per pixel (SH + [3,3] / SH + [3,4])."""
    del xnp
    if distortion_params is not None:
        raise ValueError("lens distortion is outside the fused ray generator (distortion_params must be None)")
    if camtype != ProjectionType.PERSPECTIVE:
        raise ValueError("only ProjectionType.PERSPECTIVE cameras are generated on the device")
    if device is None:
        device = pix_x_int.device if torch.is_tensor(pix_x_int) and pix_x_int.is_cuda else torch.device("cuda")
    px = _dev(pix_x_int, device, torch.int32)
    py = _dev(pix_y_int, device, torch.int32)
    sh = tuple(px.shape)
    p2c = _dev(pixtocams, device)
    c2w = _dev(camtoworlds, device)[..., :3, :4]
    if p2c.dim() > 2:
        p2c = torch.broadcast_to(p2c, sh + (3, 3)).reshape(-1, 3, 3)
    if c2w.dim() > 2:
        c2w = torch.broadcast_to(c2w, sh + (3, 4)).reshape(-1, 3, 4)
    ndc = None if pixtocam_ndc is None else _dev(pixtocam_ndc, device)
    o, d, v, r, ip = _hip.pixels_to_rays(px, py, p2c, c2w, ndc)
    return o.reshape(sh + (3,)), d.reshape(sh + (3,)), v.reshape(sh + (3,)), r.reshape(sh + (1,)), ip.reshape(sh + (2,))


def cast_ray_batch(cameras, pixels, camtype=ProjectionType.PERSPECTIVE, xnp=torch, device=None) -> utils.Rays:
    """Maps from input cameras and Pixel batch to output Ray batch (camera_utils.py:617-670)."""
    pixtocams, camtoworlds, distortion_params, pixtocam_ndc = cameras
    cam_idx = torch.as_tensor(np.asarray(pixels.cam_idx) if not torch.is_tensor(pixels.cam_idx) else pixels.cam_idx)[..., 0].long()

    def batch_index(arr):
        arr = torch.as_tensor(np.asarray(arr) if not torch.is_tensor(arr) else arr)
        return arr if arr.dim() == 2 else arr[cam_idx.to(arr.device)]
    o, d, v, r, ip = pixels_to_rays(pixels.pix_x_int, pixels.pix_y_int, batch_index(pixtocams), batch_index(camtoworlds),
                                    distortion_params=distortion_params, pixtocam_ndc=pixtocam_ndc, camtype=camtype,
                                    xnp=xnp, device=device)
    dev = o.device
    return utils.Rays(origins=o, directions=d, viewdirs=v, radii=r, imageplane=ip,
                      lossmult=_dev(pixels.lossmult, dev), near=_dev(pixels.near, dev), far=_dev(pixels.far, dev),
                      cam_idx=_dev(pixels.cam_idx, dev, torch.int32))


def cast_pinhole_rays(camtoworld, height, width, focal, near, far, xnp=torch, device=None) -> utils.Rays:
    """Pinhole camera ray batch for a whole image (camera_utils.py:673-697)."""
    device = torch.device("cuda") if device is None else device
    pix_x_int, pix_y_int = pixel_coordinates(width, height, device)
    o, d, v, r, ip = pixels_to_rays(pix_x_int, pix_y_int, get_pixtocam(focal, width, height), camtoworld, xnp=xnp, device=device)

    def broadcast_scalar(x, dtype=torch.float32):
        return torch.full(tuple(pix_x_int.shape) + (1,), x, dtype=dtype, device=device)
    return utils.Rays(origins=o, directions=d, viewdirs=v, radii=r, imageplane=ip, lossmult=broadcast_scalar(1.),
                      near=broadcast_scalar(float(near)), far=broadcast_scalar(float(far)),
                      cam_idx=broadcast_scalar(0, torch.int32))
