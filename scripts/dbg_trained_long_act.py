"""f32 vs split-f16 training forward of one level on the long-trained weights: where do the saved activations / ReLU masks differ?"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from helpers import load_golden, params_from_golden, rays_from_golden
import refnerf_pl_amd
from refnerf_pl_amd import _hip
g = load_golden("model_trained_long_train")
dev = "cuda:0"
P = torch.tensor(params_from_golden(g), device=dev)
rays = {k: torch.tensor(v, device=dev) for k, v in rays_from_golden(g).items()}
for k in ("radii", "near", "far"): rays[k] = rays[k].reshape(-1)
R = rays["origins"].shape[0]; N = 64
packed = _hip.pack_weights(P, precision=0)
sd = torch.tensor([[0.0, 1.0]], device=dev).repeat(R, 1); w = torch.ones((R, 1), device=dev)
acts = {}
for prec in (0, 3):
    cfg = _hip.default_cfg(n_samples=N, n_in=1, training=1, compute_extras=0); cfg.precision = prec
    res = _hip.level_forward(packed, cfg, rays, sd, w, history=True, save_activations=True)
    torch.cuda.synchronize()
    a = res["activations"].view(torch.float32).cpu().numpy()
    acts[prec] = (a, {k: res[k].cpu().numpy() for k in ("density", "rgb", "normals", "weights")})
S = R * N
UNITS = 4525
def row(a, u):   # blocked layout: (s>>6)*units*64 + u*64 + (s&63)
    s = np.arange(S)
    return a[(s >> 6) * UNITS * 64 + u * 64 + (s & 63)]
a0, a3 = acts[0][0], acts[3][0]
ACT_SP, ACT_MASK = 96, 4396
for L in range(8):
    rows0 = np.stack([row(a0, ACT_SP + L * 256 + j) for j in range(0, 256, 8)])
    rows3 = np.stack([row(a3, ACT_SP + L * 256 + j) for j in range(0, 256, 8)])
    d = np.abs(rows0 - rows3)
    flips = np.sum((rows0 > 0) != (rows3 > 0))
    print(f"x{L} (input of layer {L+1}): max abs diff {d.max():.2e}, max value {np.abs(rows0).max():.2f}, sign/zero flips {flips} of {rows0.size}")
for L in range(16):
    m0 = np.stack([row(a0, ACT_MASK + 8 * L + q) for q in range(8)]).view(np.uint32)
    m3 = np.stack([row(a3, ACT_MASK + 8 * L + q) for q in range(8)]).view(np.uint32)
    x = m0 ^ m3
    bits = sum(int(np.unpackbits(x.view(np.uint8)).sum()) for _ in [0])
    print(f"mask layer {L}: differing bits {bits} of {m0.size * 32}")
for k in ("density", "rgb", "normals", "weights"):
    print(k, "max abs diff", float(np.abs(acts[0][1][k] - acts[3][1][k]).max()))

# ---- backward (f32 chains) of this level from both forwards' saved buffers, same upstream gradients
from refnerf_pl_amd import layout
gen = torch.Generator().manual_seed(3)
g_rgb = (torch.randn((R, 3), generator=gen) * 1e-2).to(dev); g_w = (torch.randn((R, N), generator=gen) * 1e-3).to(dev); g_np = (torch.randn((R, N, 3), generator=gen) * 1e-3).to(dev)
grads = {}
full = {}
for prec in (0, 3):
    cfg = _hip.default_cfg(n_samples=N, n_in=1, training=1, compute_extras=0); cfg.precision = prec
    res = _hip.level_forward(packed, cfg, rays, sd, w, history=True, save_activations=True)
    full[prec] = res
    cfg.precision = 0
    out = torch.zeros(_hip.NUM_PARAMS, device=dev)
    _hip.level_backward(packed, cfg, rays, res, g_rgb, g_w, g_np, out)
    grads[prec] = out.cpu().numpy()
d = grads[3] - grads[0]
print("level-0 backward from f16x2-forward buffers vs f32-forward buffers: rel %.2e" % (np.linalg.norm(d) / np.linalg.norm(grads[0])))
for s in layout.PARAM_SPECS:
    a = slice(s.w_off, s.w_off + s.out_dim * s.in_dim)
    print("  %-16s %.2e" % (s.name, np.linalg.norm(d[a]) / max(np.linalg.norm(grads[0][a]), 1e-30)))
# which saved rows differ most (all 4396 operand rows)
a0 = full[0]["activations"].view(torch.float32).cpu().numpy(); a3 = full[3]["activations"].view(torch.float32).cpu().numpy()
worst = []
for u in range(0, 4396):
    r0, r3 = row(a0, u), row(a3, u)
    den = np.abs(r0).max()
    if den > 0: worst.append((float(np.abs(r0 - r3).max() / den), u))
worst.sort()
print("rows with the largest relative difference (rel, row):", worst[-8:])
for k in ("density", "rgb", "weights", "sdist"):
    print(k, float((full[0][k] - full[3][k]).abs().max()))

# ---- level 1 from IDENTICAL inputs (the f32 level-0 step function): forward in both modes, backward f32, same seeds
sd1, w1 = full[0]["sdist"].contiguous(), full[0]["weights"].contiguous()
N1 = 96
g_w1 = (torch.randn((R, N1), generator=gen) * 1e-3).to(dev); g_np1 = (torch.randn((R, N1, 3), generator=gen) * 1e-3).to(dev)
gr1, f1 = {}, {}
for prec in (0, 3):
    cfg = _hip.default_cfg(n_samples=N1, n_in=N, training=1, compute_extras=0); cfg.precision = prec
    res = _hip.level_forward(packed, cfg, rays, sd1, w1, history=True, save_activations=True)
    f1[prec] = res
    cfg.precision = 0
    out = torch.zeros(_hip.NUM_PARAMS, device=dev)
    _hip.level_backward(packed, cfg, rays, res, g_rgb, g_w1, g_np1, out)
    gr1[prec] = out.cpu().numpy()
d = gr1[3] - gr1[0]
print("LEVEL 1, identical step function in: sdist equal", bool((f1[0]["sdist"] == f1[3]["sdist"]).all()),
      "| density max diff %.2e" % float((f1[0]["density"] - f1[3]["density"]).abs().max()),
      "| gradient rel diff %.2e" % (np.linalg.norm(d) / np.linalg.norm(gr1[0])))
