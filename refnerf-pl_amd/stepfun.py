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


def weight_to_pdf(t, w, eps=torch.finfo(torch.float32).eps ** 2):
    """stepfun.py:92-94."""
    return w / torch.clamp(t[..., 1:] - t[..., :-1], min=eps)


def pdf_to_weight(t, p):
    """stepfun.py:97-99."""
    return p * (t[..., 1:] - t[..., :-1])


def max_dilate(t, w, dilation, domain=(-float('inf'), float('inf')), rows_per_block=512):
    """Dilate (max-pool) a non-negative step function by `dilation` on both sides (stepfun.py:102-115):
    knots = sort(t, t[:-1] - d, t[1:] + d) clipped to `domain`; the value on a new interval is the largest
    old value whose widened interval [t_i - d, t_{i+1} + d) contains its left knot.  Same comparison masks
    as the reference, built for `rows_per_block` rays at a time to bound the [rays, 3M+1, M] transient."""
    t0 = t[..., :-1] - dilation
    t1 = t[..., 1:] + dilation
    t_dilate = torch.sort(torch.cat([t, t0, t1], dim=-1), dim=-1).values
    t_dilate = torch.clamp(t_dilate, min=float(domain[0]), max=float(domain[1]))
    lead = t.shape[:-1]
    t0f, t1f, tdf, wf = (x.reshape(-1, x.shape[-1]) for x in (t0, t1, t_dilate, w))
    out = []
    for i in range(0, tdf.shape[0], rows_per_block):
        s = slice(i, i + rows_per_block)
        inside = (t0f[s, None, :] <= tdf[s, :, None]) & (t1f[s, None, :] > tdf[s, :, None])
        out.append(torch.where(inside, wf[s, None, :], torch.zeros((), dtype=wf.dtype, device=wf.device)).max(dim=-1).values[..., :-1])
    w_dilate = torch.cat(out, dim=0).reshape(lead + (t_dilate.shape[-1] - 1,))
    return t_dilate, w_dilate


def max_dilate_weights(t, w, dilation, domain=(-float('inf'), float('inf')), renormalize=False,
                       eps=torch.finfo(torch.float32).eps ** 2):
    """Dilate a set of weights through their PDF (stepfun.py:118-131)."""
    p = weight_to_pdf(t, w)
    t_dilate, p_dilate = max_dilate(t, p, dilation, domain=domain)
    w_dilate = pdf_to_weight(t_dilate, p_dilate)
    if renormalize:
        w_dilate = w_dilate / torch.clamp(torch.sum(w_dilate, dim=-1, keepdim=True), min=eps)
    return t_dilate, w_dilate
