"""Step-function helpers the losses need on the host -- mirror of the reference's internal/stepfun.py
searchsorted (:31-56), inner_outer (:67-80) and lossfun_outer (:83-89).  (The resampler of the same
file, sample_intervals / invert_cdf, is inside the fused level kernel: rn::sample_intervals_wave.)
"""
import torch


def searchsorted(a, v):
    """(idx_lo, idx_hi) with a[idx_lo] <= v < a[idx_hi]; both clamp to the first / last index of `a`
    when v lies outside [a[0], a[-1]] (stepfun.py:31-56).  Binary search instead of the reference's
    O(len(a) * len(v)) comparison masks."""
    cnt = torch.searchsorted(a.contiguous(), v.contiguous(), right=True)          # number of a_i <= v
    last = a.shape[-1] - 1
    return torch.clamp(cnt - 1, min=0), torch.clamp(cnt, max=last)


def inner_outer(t0, t1, y1):
    """Inner and outer measures of the step function (t1, y1) on the intervals of t0 (stepfun.py:67-80)."""
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    idx_lo, idx_hi = searchsorted(t1, t0)
    cy1_lo = torch.take_along_dim(cy1, idx_lo, dim=-1)
    cy1_hi = torch.take_along_dim(cy1, idx_hi, dim=-1)
    y0_outer = cy1_hi[..., 1:] - cy1_lo[..., :-1]
    y0_inner = torch.where(idx_hi[..., :-1] <= idx_lo[..., 1:], cy1_lo[..., 1:] - cy1_hi[..., :-1],
                           torch.zeros_like(y0_outer))
    return y0_inner, y0_outer


def lossfun_outer(t, w, t_env, w_env, eps=torch.finfo(torch.float32).eps):
    """The proposal weights (t_env, w_env) should be an upper envelope of the NeRF weights (t, w):
    scaled half-quadratic penalty on the excess (stepfun.py:83-89)."""
    _, w_outer = inner_outer(t, t_env, w_env)
    return torch.clamp(w - w_outer, min=0.0) ** 2 / (w + eps)
