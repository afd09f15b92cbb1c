"""Ray containers: mirror of the reference's ``internal/utils.py:30-118``."""
from dataclasses import dataclass, fields
from typing import Optional, Union

import numpy as np
import torch

_Array = Union[np.ndarray, torch.Tensor]


@dataclass
class Rays:
    """All tensors share leading dims; last dim is the channel (utils.py:51-93)."""
    origins: _Array
    directions: _Array
    viewdirs: _Array
    radii: _Array
    imageplane: _Array
    lossmult: _Array
    near: _Array
    far: _Array
    cam_idx: _Array

    def __getitem__(self, s):
        if isinstance(s, int):
            return Rays(*[[getattr(self, f.name)[s]] for f in fields(self)])
        elif isinstance(s, slice):
            return Rays(*[getattr(self, f.name)[s] for f in fields(self)])
        raise ValueError('Argument to __getitem__ must be int or slice')

    def to(self, device):
        """In-place: numpy fields become float32 tensors on `device` (utils.py:74-85).
        Unlike the reference (whose tensor branch discards the moved copy), tensor
        fields already on another device ARE moved."""
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, np.ndarray):
                setattr(self, f.name, torch.tensor(v, dtype=torch.float32, device=device))
            elif isinstance(v, torch.Tensor):
                if v.device != torch.device(device):
                    setattr(self, f.name, v.to(device))
            else:
                raise ValueError('Rays members must be either np.ndarray or torch.Tensor')
        return self

    def reshape(self, *dims):
        return Rays(*[getattr(self, f.name).reshape(*dims) for f in fields(self)])

    @property
    def shape(self):
        return self.origins.shape


def dummy_rays() -> Rays:
    """utils.py:96-107"""
    def data_fn(n):
        return torch.zeros((1, n))
    return Rays(origins=data_fn(3), directions=data_fn(3), viewdirs=data_fn(3), radii=data_fn(1),
                imageplane=data_fn(2), lossmult=data_fn(1), near=data_fn(1), far=data_fn(1),
                cam_idx=data_fn(1).type(torch.int32))


def rays_from_dict(d: dict, device=None) -> Rays:
    r = Rays(**{f.name: d[f.name] for f in fields(Rays)})
    if device is not None:
        r.to(device)
    return r


@dataclass
class Pixels:
    """utils.py:31-48: integer pixel coordinates + per-pixel metadata, any batch shape."""
    pix_x_int: _Array
    pix_y_int: _Array
    lossmult: _Array
    near: _Array
    far: _Array
    cam_idx: _Array


@dataclass
class Batch:
    """utils.py:110-118"""
    rays: Rays
    rgb: Optional[_Array] = None
    disps: Optional[_Array] = None
    normals: Optional[_Array] = None
    alphas: Optional[_Array] = None


def merge_chunks(chunks):
    """utils.py:192-204"""
    merged = {}
    for key in chunks[0]:
        if isinstance(chunks[0][key], list):
            merged[key] = [torch.cat([c[key][i] for c in chunks]) for i in range(len(chunks[0][key]))]
        elif isinstance(chunks[0][key], torch.Tensor):
            merged[key] = torch.cat([c[key] for c in chunks])
        else:
            raise ValueError('Contents should be either list or tensor')
    return merged


def recursive_detach(v):
    if isinstance(v, torch.Tensor):
        return v.detach()
    elif isinstance(v, list):
        return [recursive_detach(x) for x in v]
    raise ValueError('Contents should be either list or tensor')


# ------------------------------------------------------------------------------------------------
# Data formats either side of the path: the reference's image artefacts and its
# Lightning checkpoints.

def _to_numpy(x):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def load_img(pth: str) -> np.ndarray:
    """utils.py:162-166: any PIL-readable image as float32."""
    from PIL import Image
    with open(pth, 'rb') as f:
        return np.array(Image.open(f), dtype=np.float32)


def save_img_u8(img, pth, mask=None):
    """utils.py:169-182: image in [0, 1] -> uint8 PNG (NaN -> 0, clipped).  With `mask` (the reference
    uses it for the roughness map matted by `acc`) the image is added to the inverted min-max normalised
    mask and the sum is min-max normalised again."""
    from PIL import Image
    px = (np.clip(np.nan_to_num(_to_numpy(img)), 0., 1.) * 255).astype(np.uint8).squeeze()
    if mask is not None:
        m = np.nan_to_num(_to_numpy(mask)).astype(np.float32).squeeze()
        m = 255 * (m - m.min()) / (m.max() - m.min())
        s = (255 - m) + px
        px = np.array(255 * (s - s.min()) / (s.max() - s.min()), dtype=np.uint8)
    with open(pth, 'wb') as f:
        Image.fromarray(px).save(f, 'PNG')


def save_img_f32(depthmap, pth):
    """utils.py:185-189: float32 TIFF (NaN -> 0)."""
    from PIL import Image
    with open(pth, 'wb') as f:
        Image.fromarray(np.nan_to_num(_to_numpy(depthmap)).astype(np.float32)).save(f, 'TIFF')


def write_render_outputs(rendering, out_dir: str, idx) -> list:
    """The artefact set of NeRFSystem.test_step (nerf_system.py:506-531) for one rendered view:
    color_ / diffuse_ / specular_ / normals_pred_ / rho_ PNGs, distance_mean_ / distance_median_ /
    acc_ float TIFFs, named with the zero-padded view index.  Returns the paths written."""
    import os
    idx_str = idx if isinstance(idx, str) else f'{idx:03d}'
    r = {k: _to_numpy(v).astype(np.float64) for k, v in rendering.items()
         if k in ('rgb', 'diffuse', 'specular', 'normals_pred', 'acc', 'distance_mean', 'distance_median', 'roughness')}
    os.makedirs(out_dir, exist_ok=True)
    done = []

    def out(fn, x, name, **kw):
        pth = os.path.join(out_dir, name)
        fn(x, pth, **kw)
        done.append(pth)
    out(save_img_u8, r['rgb'], f'color_{idx_str}.png')
    out(save_img_u8, r['diffuse'], f'diffuse_{idx_str}.png')
    out(save_img_u8, r['specular'], f'specular_{idx_str}.png')
    if 'normals_pred' in r:
        out(save_img_u8, r['normals_pred'] / 2. + 0.5, f'normals_pred_{idx_str}.png')
    out(save_img_f32, r['distance_mean'], f'distance_mean_{idx_str}.tiff')
    out(save_img_f32, r['distance_median'], f'distance_median_{idx_str}.tiff')
    out(save_img_f32, r['acc'], f'acc_{idx_str}.tiff')
    out(save_img_u8, r['roughness'], f'rho_{idx_str}.png', mask=r['acc'])
    return done


CKPT_PREFIX = 'model.'       # NeRFSystem holds the Model as `self.model` (nerf_system.py:22-33)


def load_reference_checkpoint(model, ckpt, strict: bool = True):
    """Load a checkpoint written by the reference's Lightning system into `model` (refnerf_pl_amd.models.Model).

    `ckpt`: path of a `.ckpt` file, the dict torch.load returns for one, or a bare state_dict.  Keys are the
    reference's (`model.nerf_mlp.spatial_net.0.weight`, ... and the same 46 tensors again under
    `model.prop_mlp.*` when Model.single_mlp); the `model.` prefix is optional.  Returns the
    (missing, unexpected) key lists of load_state_dict."""
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, '__fspath__'):
        ckpt = torch.load(ckpt, map_location='cpu', weights_only=False)
    sd = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
    sd = {(k[len(CKPT_PREFIX):] if k.startswith(CKPT_PREFIX) else k): v for k, v in sd.items()}
    res = model.load_state_dict(sd, strict=strict)
    mlps = {id(m): m for m in (model.nerf_mlp, model.prop_mlp)}
    for mlp in mlps.values():                  # the fused path reads the canonical blob: re-alias it
        mlp._flat = None
        mlp.flat_params()
    return list(res.missing_keys), list(res.unexpected_keys)


def reference_checkpoint(model, **extra) -> dict:
    """The Lightning-shaped dict for `model` (`state_dict` with the `model.` prefix) -- what the
    reference's `NeRFSystem.load_from_checkpoint` / `trainer.fit(ckpt_path=...)` read."""
    sd = {CKPT_PREFIX + k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    return dict(state_dict=sd, **extra)
