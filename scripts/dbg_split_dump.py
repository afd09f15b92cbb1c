"""Debug (GPU): per-layer check of the split-f16 kernel's spatial trunk.  Builds a -DREFNERF_SPLIT_DUMP library, runs ONE
level on the trained-like weights, reads back what the kernel saw (IPE features as hi + lo) and produced (every spatial
layer's ReLU output, the scalar head rows) and checks each layer in float64 against the kernel's OWN previous layer:
isolates the layer / feature / sample where an error is injected."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
CS = os.path.join(ROOT, "refnerf-pl_amd", "csrc")
lib = os.path.join(ROOT, "gpurun_out", "librefnerf_hip_dump.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
                       "-DREFNERF_SPLIT_DUMP", "-I", os.path.join(ROOT, "include"), "-I", CS, os.path.join(CS, "refnerf_hip.hip"), "-o", lib])
import refnerf_pl_amd
from refnerf_pl_amd import _hip, synthetic, layout
_hip.LIB_PATH = lib
from test_hip_parity import run_hip_model
from helpers import trained_blob
P = trained_blob()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rays = synthetic.blender_rays(4096, seed=3, center_frac=0.8)
rays = {k: v[:R] for k, v in rays.items()}
N = 128
STRIDE = 96 + 8 * 256 + 16
buf = torch.zeros((R * N, STRIDE), device="cuda")
L = _hip.lib()
L.refnerf_debug_set_dump.argtypes = [C.c_void_p]
L.refnerf_debug_set_dump(C.c_void_p(buf.data_ptr()))
lv = dict(num_levels=1, num_nerf_samples=N)
a = run_hip_model(_hip, P, rays, {}, lv, precision=3)[0]
torch.cuda.synchronize()
D = buf.cpu().numpy().astype(np.float64)
L.refnerf_debug_set_dump(None)
b = run_hip_model(_hip, P, rays, {}, lv, precision=0)[0]
rel = (np.abs(a["density"] - b["density"]) / np.maximum(np.abs(b["density"]), 1e-3)).reshape(-1)
print("bad samples (density rel err > 2e-4):", int((rel > 2e-4).sum()), "of", rel.size)
W = {}
for spec in layout.PARAM_SPECS:
    n = spec.out_dim * spec.in_dim
    W[spec.name] = (P[spec.w_off:spec.w_off + n].reshape(spec.out_dim, spec.in_dim).astype(np.float64), P[spec.b_off:spec.b_off + spec.out_dim].astype(np.float64))
feat = D[:, :96]
prev = feat
for i in range(8):
    w, bias = W["spatial_net.%d" % i]
    x = np.concatenate([prev, feat], 1) if i == 5 else prev
    exp = np.maximum(x @ w.T + bias, 0.0)
    mag = np.abs(x) @ np.abs(w.T) + np.abs(bias)
    got = D[:, 96 + 256 * i: 96 + 256 * (i + 1)]
    err = np.abs(got - exp) / mag
    worst = np.unravel_index(err.argmax(), err.shape)
    nbad = int((err > 2e-6).sum())
    print("layer %d: worst |err| / sum|products| = %.2e at sample %d feature %d (got %.6g expected %.6g); entries > 2e-6: %d; samples touched: %d" % (
        i, err.max(), worst[0], worst[1], got[worst], exp[worst], nbad, int((err > 2e-6).any(1).sum())))
    if nbad:
        bs, bf_ = np.nonzero(err > 2e-6)
        print("   first bad (sample, sample%%32, feature, got, expected):", [(int(s_), int(s_ % 32), int(f_), float(got[s_, f_]), float(exp[s_, f_])) for s_, f_ in list(zip(bs, bf_))[:12]])
        print("   bad samples that are also density-bad:", int((rel[np.unique(bs)] > 2e-4).sum()), "of", len(np.unique(bs)))
    prev = got

# ---- where do the layer-0-bad samples sit?  (rpw = 4 rays x 128 = 512 samples per workgroup, 2 passes of 256)
w0, b0 = W["spatial_net.0"]
exp0 = np.maximum(feat @ w0.T + b0, 0.0)
mag0 = np.abs(feat) @ np.abs(w0.T) + np.abs(b0)
got0 = D[:, 96:96 + 256]
bad = np.nonzero((np.abs(got0 - exp0) / mag0 > 2e-6).any(1))[0]
g = bad % 512
for name, v, nb in (("pass", g // 256, 2), ("wave", (g % 256) // 32, 8), ("run", (g % 32) // 16, 2), ("i16", g % 16, 16), ("ray in wg", g // 128, 4)):
    print(name, np.bincount(v, minlength=nb))
print("workgroups with bad samples:", len(np.unique(bad // 512)), "of", R // 4, "; bad per affected wg (max):", np.bincount(bad // 512).max())
# which input features explain the error?  least squares on the active outputs of a few bad samples
for s_ in bad[:6]:
    act = exp0[s_] > 0
    dx, *_ = np.linalg.lstsq(w0[act], (got0[s_] - exp0[s_])[act], rcond=None)
    top = np.argsort(-np.abs(dx))[:6]
    print("sample", int(s_), "g", int(s_ % 512), "top feature deltas:", [(int(k), "%.2e" % dx[k], "feat %.3e" % feat[s_, k]) for k in top])
