"""Unfused PyTorch-CPU restatement of the Ref-NeRF level loop (eval forward).

TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything under oracle/): bench.py's cpu_baseline leg times it as
"how the reference evaluates the path" -- one ATen op after another on [R, N, ...] tensors on the host cores --
and tests/test_oracle_golden.py checks it against the C oracle.  Written from the numerical spec in SURVEY.md
Appendix A (which cites the reference lines: stepfun.py:134-258, coord.py:63-126, render.py:22-254,
ref_utils.py:22-161, models.py:533-750); no reference code is imported or copied, and the product package never
imports this module.  Parity status: checked against oracle/refnerf_oracle.c (itself pinned by tests/golden).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPS = float(torch.finfo(torch.float32).eps)
WIDTH, DEPTH, IPE_DEG, BNECK = 256, 8, 16, 128
IDE_LS = (1, 2, 4, 8, 16)


def unpack(blob, specs=None):
    """canonical fp32 blob (or, with `specs`, a variant's own flat blob: layout.variant_layout) -> {name: (weight [out,in],
    bias [out])} as torch tensors."""
    from refnerf_pl_amd import layout
    blob = torch.as_tensor(np.asarray(blob, np.float32))
    out = {}
    for spec in (specs or layout.PARAM_SPECS):
        n = spec.out_dim * spec.in_dim
        out[spec.name] = (blob[spec.w_off:spec.w_off + n].reshape(spec.out_dim, spec.in_dim),
                          blob[spec.b_off:spec.b_off + spec.out_dim])
    return out


def _ide_tables():
    """Monomial coefficients of the 36 (l, m) terms, l in {1,2,4,8,16}, m = 0..l, float64 -> float32:
    Y_lm(x,y,z) = (x+iy)^m * sum_k z^k mat[k, (l,m)]."""
    def gbinom(a, k):
        return np.prod(a - np.arange(k)) / math.factorial(k)
    pairs = [(l, m) for l in IDE_LS for m in range(l + 1)]
    mat = np.zeros((IDE_LS[-1] + 1, len(pairs)))
    for col, (l, m) in enumerate(pairs):
        norm = math.sqrt((2 * l + 1) * math.factorial(l - m) / (4 * math.pi * math.factorial(l + m)))
        for k in range(l - m + 1):
            leg = (-1) ** m * 2 ** l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m) * \
                gbinom(0.5 * (l + k + m - 1.0), l)
            mat[k, col] = norm * leg
    ms = torch.tensor([m for _, m in pairs])
    sig = torch.tensor([0.5 * l * (l + 1) for l, _ in pairs], dtype=torch.float32)
    return torch.tensor(mat, dtype=torch.float32), ms, sig


_IDE = None


def integrated_dir_enc(xyz, kappa_inv):
    """[.., 3], [.., 1] -> [.., 72] = [Re x 36 | Im x 36]."""
    global _IDE
    if _IDE is None:
        _IDE = _ide_tables()
    mat, ms, sig = _IDE
    x, y, z = xyz[..., 0:1], xyz[..., 1:2], xyz[..., 2:3]
    zp = torch.cat([z ** k for k in range(mat.shape[0])], dim=-1)
    pr, pi = [torch.ones_like(x)], [torch.zeros_like(x)]
    for _ in range(IDE_LS[-1]):
        pr.append(pr[-1] * x - pi[-1] * y)
        pi.append(pr[-2] * y + pi[-1] * x)
    pr, pi = torch.cat(pr, -1)[..., ms], torch.cat(pi, -1)[..., ms]
    poly = zp @ mat
    att = torch.exp(-sig * kappa_inv)
    return torch.cat([pr * poly * att, pi * poly * att], dim=-1)


def linear_to_srgb(x):
    return torch.where(x <= 0.0031308, (323.0 / 25.0) * x, (211.0 * torch.clamp(x, min=EPS) ** (5.0 / 12.0) - 11.0) / 200.0)


def sample_intervals(t, w, n, anneal=1.0, padding=0.01, s_near=0.0, s_far=1.0):
    """SURVEY.md A1 with a sorted search instead of the reference's [R, M+1, N] masks."""
    logits = torch.where(t[..., 1:] > t[..., :-1], anneal * torch.log(w + padding), torch.full_like(w, -float("inf")))
    p = torch.softmax(logits, dim=-1)
    cw = torch.cat([torch.zeros_like(p[..., :1]), torch.clamp(torch.cumsum(p[..., :-1], dim=-1), max=1.0),
                    torch.ones_like(p[..., :1])], dim=-1)
    pad = 1.0 / (2 * n)
    u = torch.linspace(pad, 1.0 - pad - EPS, n).expand(t.shape[0], n).contiguous()
    M = w.shape[-1]
    i0 = torch.clamp(torch.searchsorted(cw, u, right=True) - 1, 0, M)
    i1 = torch.clamp(i0 + 1, max=M)
    xp0, xp1 = torch.gather(cw, -1, i0), torch.gather(cw, -1, i1)
    fp0, fp1 = torch.gather(t, -1, i0), torch.gather(t, -1, i1)
    off = torch.clamp(torch.nan_to_num((u - xp0) / (xp1 - xp0), 0.0), 0.0, 1.0)
    c = fp0 + off * (fp1 - fp0)
    mid = (c[..., 1:] + c[..., :-1]) / 2
    first = torch.clamp(2 * c[..., :1] - mid[..., :1], min=s_near)
    last = torch.clamp(2 * c[..., -1:] - mid[..., -1:], max=s_far)
    return torch.cat([first, mid, last], dim=-1), i0


def level_forward(P, rays, sdist_in, weights_in, n_samples, srgb_mapping=True, render_srgb_mode="none",
                  density_bias=0.5, roughness_bias=-1.0, rgb_padding=0.001, bg_rgb=1.0, basis=None):
    """`basis` [n, 3]: the unit directions of the IPE basis as rows (geopoly.generate_basis; None = octahedron / 1)"""
    o, d, v = rays["origins"], rays["directions"], rays["viewdirs"]
    radii, near, far = rays["radii"].reshape(-1, 1), rays["near"].reshape(-1, 1), rays["far"].reshape(-1, 1)
    sdist, bin_idx = sample_intervals(sdist_in, weights_in, n_samples)
    tdist = sdist * far + (1 - sdist) * near                                            # A2
    t0, t1 = tdist[..., :-1], tdist[..., 1:]                                            # A3
    mu, hw = (t0 + t1) / 2, (t1 - t0) / 2
    den = torch.clamp(3 * mu ** 2 + hw ** 2, min=EPS)
    t_mean = mu + (2 * mu * hw ** 2) / den
    t_var = hw ** 2 / 3 - (4 / 15) * (hw ** 4 * (12 * mu ** 2 - hw ** 2)) / den ** 2
    r_var = (mu ** 2 / 4 + (5 / 12) * hw ** 2 - (4 / 15) * hw ** 4 / den) * radii ** 2
    mean = o[:, None, :] + d[:, None, :] * t_mean[..., None]
    d2 = torch.clamp((d * d).sum(-1, keepdim=True), min=1e-10)
    if basis is None:
        basis = torch.tensor([[0.0, 0.0, -1.0], [0.0, -1.0, 0.0], [-1.0, 0.0, 0.0]])   # octahedron / 1 subdivision (symmetric)
    else:
        basis = torch.as_tensor(np.asarray(basis, np.float32)).T                        # [3, n]: columns = directions
    nb = basis.shape[1]
    lmean = mean @ basis
    bd = d @ basis                                                                       # b . d per basis column
    lvar = t_var[..., None] * (bd ** 2)[:, None, :] + r_var[..., None] * (1.0 - (bd * (bd / d2))[:, None, :])
    scales = 2.0 ** torch.arange(IPE_DEG, dtype=torch.float32)                           # A4
    x = (lmean[..., None, :] * scales[:, None]).reshape(lmean.shape[:-1] + (IPE_DEG * nb,))
    s = (lvar[..., None, :] * scales[:, None] ** 2).reshape(lvar.shape[:-1] + (IPE_DEG * nb,))

    def safe_sin(a):
        return torch.sin(torch.where(a.abs() < 100 * math.pi, a, a % (100 * math.pi)))
    feat = torch.cat([torch.exp(-0.5 * s) * safe_sin(x), torch.exp(-0.5 * s) * safe_sin(x + 0.5 * math.pi)], dim=-1)
    h = feat                                                                             # A5
    for i in range(DEPTH):
        h = F.relu(F.linear(h, *P["spatial_net.%d" % i]))
        if i == 4:
            h = torch.cat([h, feat], dim=-1)
    raw_density = F.linear(h, *P["raw_density"])[..., 0]
    gp = F.linear(h, *P["grad_pred"])
    npred = -gp / torch.sqrt(torch.clamp((gp ** 2).sum(-1, keepdim=True), min=EPS))
    rough = F.softplus(F.linear(h, *P["raw_roughness"]) + roughness_bias)
    raw_dif = F.linear(h, *P["raw_rgb_diffuse"])
    tint = torch.sigmoid(F.linear(h, *P["raw_tint"]))
    bneck = F.linear(h, *P["bottleneck"])
    density = F.softplus(raw_density + density_bias)
    w3 = -v[:, None, :]
    refdir = 2.0 * (npred * w3).sum(-1, keepdim=True) * npred - w3
    dotv = (npred * v[:, None, :]).sum(-1, keepdim=True)
    din = torch.cat([bneck, integrated_dir_enc(refdir, rough), dotv], dim=-1)
    h = din
    for i in range(DEPTH):
        h = F.relu(F.linear(h, *P["viewdir_mlp.%d" % i]))
        if i == 4:
            h = torch.cat([h, din], dim=-1)
    spec_lin = tint * torch.sigmoid(F.linear(h, *P["rgb"]))
    dif_lin = torch.sigmoid(raw_dif - math.log(3.0))
    rgb = spec_lin + dif_lin
    if srgb_mapping:
        rgb = rgb / torch.clamp(rgb.max(dim=-1, keepdim=True).values, min=1.0)
        rgb, dif, spc = (torch.clamp(linear_to_srgb(c), 0.0, 1.0) for c in (rgb, dif_lin, spec_lin))
    else:
        dif, spc = dif_lin, spec_lin
    rgb = rgb * (1 + 2 * rgb_padding) - rgb_padding
    delta = (t1 - t0) * torch.linalg.norm(d, dim=-1, keepdim=True)                       # A7
    dd = density * delta
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], dim=-1)], dim=-1))
    weights = (1 - torch.exp(-dd)) * trans
    acc = weights.sum(-1)
    bg_w = torch.clamp(1 - acc, min=0.0)
    comp = {k: (weights[..., None] * c).sum(-2) + bg_w[..., None] * bg_rgb for k, c in (("rgb", rgb), ("diffuse", dif), ("specular", spc))}
    if render_srgb_mode != "none":
        if render_srgb_mode.startswith("norm"):
            comp["rgb"] = comp["rgb"] / torch.clamp(comp["rgb"].max(dim=-1, keepdim=True).values, min=1.0)
        if render_srgb_mode.endswith("srgb"):
            comp = {k: linear_to_srgb(c) for k, c in comp.items()}
        comp = {k: torch.clamp(c, 0.0, 1.0) for k, c in comp.items()}
    t_mid = 0.5 * (t0 + t1)
    out = {"sdist": sdist, "bin_idx": bin_idx, "weights": weights, "density": density, "rgb": rgb, "roughness": rough[..., 0],
           "normals_pred": npred, "r_rgb": comp["rgb"], "r_diffuse": comp["diffuse"], "r_specular": comp["specular"],
           "r_acc": acc, "r_distance": (weights * t_mid).sum(-1),
           "r_normals_pred": (weights[..., None] * npred).sum(-2), "r_tint": (weights[..., None] * tint).sum(-2),
           "r_roughness": (weights * rough[..., 0]).sum(-1)}
    dm = torch.exp((weights * torch.log(t_mid)).sum(-1) / torch.clamp(acc, min=EPS))
    out["r_distance_mean"] = torch.minimum(torch.maximum(torch.nan_to_num(dm, float("inf")), tdist[..., 0]), tdist[..., -1])
    # float64 percentiles of ([tdist, far], [w, bg_w]) (stepfun.py:294-307, math.py:114-142)
    cwp = torch.cat([torch.zeros_like(acc[..., None]), torch.clamp(torch.cumsum(weights, dim=-1), max=1.0),
                     torch.ones_like(acc[..., None])], dim=-1).double()
    fp = torch.cat([tdist, far], dim=-1).double()
    ps = torch.tensor([5.0, 50.0, 95.0], dtype=torch.float32).div(100.0).double().expand(acc.shape[0], 3).contiguous()
    idx = torch.clamp(torch.searchsorted(cwp, ps, right=True) - 1, 0, cwp.shape[-1] - 2)
    x0, x1 = torch.gather(cwp, -1, idx), torch.gather(cwp, -1, idx + 1)
    f0, f1 = torch.gather(fp, -1, idx), torch.gather(fp, -1, idx + 1)
    m = (f1 - f0) / (x1 - x0)
    out["r_percentiles"] = m * ps + (f0 - m * x0)
    return out


def model_forward(blob, rays, num_levels=2, num_prop_samples=128, num_nerf_samples=128, specs=None, **kw):
    """Model.__call__ eval forward -> list of per-level dicts of numpy arrays (names of oracle.model_forward)."""
    P = unpack(blob, specs)
    r = {k: torch.as_tensor(np.asarray(v, np.float32)) for k, v in rays.items()}
    R = r["origins"].shape[0]
    sdist = torch.tensor([[0.0, 1.0]]).repeat(R, 1)
    weights = torch.ones((R, 1))
    outs = []
    with torch.no_grad():
        for lvl in range(num_levels):
            n = num_prop_samples if lvl < num_levels - 1 else num_nerf_samples
            res = level_forward(P, r, sdist, weights, n, **kw)
            sdist, weights = res["sdist"], res["weights"]
            outs.append({k: v.numpy() for k, v in res.items()})
    return outs


if __name__ == "__main__":
    # python -m oracle.torch_path <blender|llff> <rays> <samples> <levels> '<make_params kwargs json>': one JSON line with
    # the best eval-forward rate over up to three intra-op thread counts (16 / 32 / 64) -- bench.py's cpu_baseline_torch
    import json
    import os
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import refnerf_pl_amd  # noqa: F401
    from refnerf_pl_amd import synthetic
    family, n_rays, N, levels = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    blob = synthetic.make_params(**json.loads(sys.argv[5]))
    rays = synthetic.llff_rays(n_rays, seed=1) if family == "llff" else synthetic.blender_rays(n_rays, seed=1, center_frac=0.5)
    cores = os.cpu_count() or 1
    best = None
    # intra-op thread counts: ATen's elementwise ops and small GEMMs get slower, not faster, with hundreds of threads
    # (256 threads: > 50 s per pass at C1's shape on the 256-core GPU host, 32 threads: 0.7 s)
    counts = sorted({min(cores, 16), min(cores, 32), min(cores, 64)})
    small = {k: v[:64] for k, v in rays.items()}
    tried = 0
    for th in counts:
        torch.set_num_threads(th)
        model_forward(blob, small, num_levels=levels, num_prop_samples=N, num_nerf_samples=N)      # warm-up (thread pool, tables)
        t0 = time.time()
        model_forward(blob, rays, num_levels=levels, num_prop_samples=N, num_nerf_samples=N)
        dt = time.time() - t0
        tried += 1
        if best is None or dt < best[0]:
            best = (dt, th)
        if dt > 20:
            break
    print(json.dumps({"rate": n_rays * N * levels / best[0], "seconds": best[0], "threads": best[1], "tried": tried}))
