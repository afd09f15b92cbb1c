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
