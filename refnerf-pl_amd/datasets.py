"""The random-ray training batcher in front of the hot path, on the device (second half).

Mirrors `Dataset._next_train` / `Dataset._make_ray_batch` of the reference's internal/datasets.py (:395-485) for
perspective cameras with `Config.cast_rays_in_train_step` semantics (configs.py:80): a batch is `batch_size //
patch_size**2` random pixel patches drawn from all images ('all_images') or from one random image ('single_image');
the colours are gathered and the rays cast (`camera_utils.cast_ray_batch` -> `refnerf_pixels_to_rays`) on the device, so
a training step moves no per-ray data over PCIe.  Everything outside that (file loaders, pose normalisation, test / path
cameras) stays out of scope.

Random numbers: a `torch.Generator` on the device, seeded per rank (`seed + rank`): the reference draws from the global
`numpy.random` state, which every DDP rank and DataLoader worker inherits identically -- not
reproduced on purpose.
"""
from typing import Optional, Sequence

import numpy as np
import torch

from . import camera_utils, utils


class TrainRayBatcher:
    """images [n, H, W, C] (device or host, float), cameras = (pixtocams [n,3,3] | [3,3], camtoworlds [n,3,4], None,
    pixtocam_ndc | None) -- the tuple `Dataset.cameras` holds in the reference."""

    def __init__(self, images, cameras: Sequence, near: float, far: float, batch_size: int, patch_size: int = 1,
                 batching: str = 'all_images', seed: int = 0, rank: int = 0, device: Optional[torch.device] = None,
                 debug_mode: bool = False):
        if batching not in ('all_images', 'single_image'):
            raise ValueError(f'Unknown batching method {batching}')
        self.device = torch.device('cuda') if device is None else torch.device(device)
        self.images = torch.as_tensor(np.asarray(images) if not torch.is_tensor(images) else images, dtype=torch.float32).to(self.device)
        self.n_examples, self.height, self.width = self.images.shape[:3]
        pixtocams, camtoworlds, distortion, ndc = cameras
        if distortion is not None:
            raise ValueError('lens distortion is outside the device ray generator')
        f32 = dict(dtype=torch.float32, device=self.device)
        self.cameras = (torch.as_tensor(np.asarray(pixtocams), **f32), torch.as_tensor(np.asarray(camtoworlds), **f32)[..., :3, :4],
                        None, None if ndc is None else torch.as_tensor(np.asarray(ndc), **f32))
        self.near, self.far = float(near), float(far)
        self.batch_size, self.patch_size, self.batching, self.debug_mode = int(batch_size), int(patch_size), batching, debug_mode
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed) + int(rank))

    def sample_pixels(self) -> utils.Pixels:
        """datasets.py:449-485: pixel patches + camera indices of one batch, shape [num_patches, patch, patch]."""
        num_patches = self.batch_size // self.patch_size ** 2
        upper = self.patch_size - 1
        dev = self.device
        if self.debug_mode:
            xs = torch.arange(0, self.width - upper, device=dev)
            ys = torch.arange(0, self.height - upper, device=dev)
            gx, gy = torch.meshgrid(xs, ys, indexing='xy')
            px = gx.reshape(-1)[:num_patches].reshape(-1, 1, 1)
            py = gy.reshape(-1)[:num_patches].reshape(-1, 1, 1)
            cam = torch.zeros((num_patches, 1, 1), dtype=torch.int64, device=dev)
        else:
            px = torch.randint(0, self.width - upper, (num_patches, 1, 1), generator=self.gen, device=dev)
            py = torch.randint(0, self.height - upper, (num_patches, 1, 1), generator=self.gen, device=dev)
            if self.batching == 'all_images':
                cam = torch.randint(0, self.n_examples, (num_patches, 1, 1), generator=self.gen, device=dev)
            else:
                cam = torch.randint(0, self.n_examples, (1,), generator=self.gen, device=dev).reshape(1, 1, 1)
        if not self.debug_mode:                            # the reference adds the patch offsets in the random branch only (datasets.py:459-476)
            d = torch.arange(self.patch_size, device=dev)
            px = px + d.reshape(1, 1, -1)                  # patch offsets (camera_utils.pixel_coordinates)
            py = py + d.reshape(1, -1, 1)
        px, py, cam = torch.broadcast_tensors(px, py, cam)
        shape = tuple(px.shape)

        def scalar(x):
            return torch.full(shape + (1,), x, dtype=torch.float32, device=dev)
        return utils.Pixels(pix_x_int=px.to(torch.int32).contiguous(), pix_y_int=py.to(torch.int32).contiguous(), lossmult=scalar(1.),
                            near=scalar(self.near), far=scalar(self.far), cam_idx=cam.to(torch.int32)[..., None].contiguous())

    def next(self, cast_rays: bool = True) -> utils.Batch:
        """One training batch: `Batch(rays=Rays | Pixels, rgb=[..., C])`, all on the device."""
        pixels = self.sample_pixels()
        cam, py, px = pixels.cam_idx[..., 0].long(), pixels.pix_y_int.long(), pixels.pix_x_int.long()
        rgb = self.images[cam, py, px]
        rays = camera_utils.cast_ray_batch(self.cameras, pixels, device=self.device) if cast_rays else pixels
        return utils.Batch(rays=rays, rgb=rgb)

    __next__ = next

    def __iter__(self):
        return self
