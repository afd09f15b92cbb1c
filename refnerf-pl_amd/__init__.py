"""refnerf-pl_amd: MI355X-native Ref-NeRF rendering inner loop.

Host-side mirror of the reference's ``internal/models.py`` call surface
(``Model``, ``MLP``, ``NerfMLP``, ``PropMLP``, ``construct_model``,
``render_image``) over a C-ABI HIP library (include/refnerf_hip.h).
"""
from . import layout  # noqa: F401

__all__ = ["layout"]
