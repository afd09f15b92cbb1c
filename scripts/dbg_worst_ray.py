"""GPU debug aid: the worst ray of a full-size f16x2 parity case (tests/test_hip_f16x2.py) -- per-sample differences of the
f16x2 mode and of the f32 mode against the oracle on that ray, with level 1 also fed the ORACLE's step function (so that the
MLP's share and the resampler's share of the error separate).  Usage: python scripts/dbg_worst_ray.py [view_seed] [R] [N].
DEBUG INFRASTRUCTURE: never imported by the product."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refnerf_pl_amd  # noqa: F401,E402
from refnerf_pl_amd import _hip as hip, synthetic  # noqa: E402
from helpers import trained_long_blob  # noqa: E402
from test_hip_parity import run_hip_model, dev_rays, DEV  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    view = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 192
    P = trained_long_blob()
    rays = synthetic.blender_rays(R, seed=view, center_frac=0.8)
    lv = dict(num_prop_samples=N, num_nerf_samples=N)
    out = run_hip_model(hip, P, rays, {}, lv, precision=3)
    f32 = run_hip_model(hip, P, rays, {}, lv, precision=0)
    d = np.abs(out[1]["r_rgb"] - f32[1]["r_rgb"]).max(-1)
    worst = np.argsort(-d)[:4]
    print("worst rays vs f32 mode:", worst, d[worst])
    for w in worst[:2]:
        one = {k: v[w:w + 1] for k, v in rays.items()}
        ref = O.model_forward(P, one, **lv)
        for nm, o in (("f16x2", out), ("f32", f32)):
            print(f"ray {w} {nm}: rgb err vs oracle L0 {np.abs(o[0]['r_rgb'][w] - ref[0]['r_rgb'][0]).max():.3e}  L1 {np.abs(o[1]['r_rgb'][w] - ref[1]['r_rgb'][0]).max():.3e}"
                  f"  sdist diff L1 {np.abs(o[1]['sdist'][w] - ref[1]['sdist'][0]).max():.3e}  bins differing {(o[1]['bin_idx'][w] != ref[1]['bin_idx'][0]).sum()}")
        # level 1 on the oracle's own step function (sdist / weights of level 0 of the oracle)
        r1 = dev_rays(one)
        for prec in (3, 0):
            packed = hip.pack_weights(torch.tensor(P, device=DEV), precision=hip.level_image(prec, False, 0))
            cfg = hip.default_cfg(n_samples=N, n_in=N, precision=prec)
            res = hip.level_forward(packed, cfg, r1, torch.tensor(ref[0]["sdist"], device=DEV), torch.tensor(ref[0]["weights"], device=DEV))
            res = {k: v.cpu().numpy() for k, v in res.items()}
            e = np.abs(res["r_rgb"][0] - ref[1]["r_rgb"][0]).max()
            print(f"   level 1 fed the oracle's step function, precision {prec}: rgb err {e:.3e}  sdist diff {np.abs(res['sdist'][0] - ref[1]['sdist'][0]).max():.3e}")
            wq, wr = res["weights"][0], ref[1]["weights"][0]
            top = np.argsort(-wr)[:6]
            for k in ("density", "rgb", "normals", "weights", "roughness"):
                if k in res and k in ref[1]:
                    a, b = res[k][0], ref[1][k][0]
                    dd = np.abs(a - b)
                    dd = dd.max(-1) if dd.ndim > 1 else dd
                    print(f"      {k:10s} max diff {dd.max():.3e} at sample {dd.argmax()} (ref there {np.ravel(b[dd.argmax()])[:3]}, weight there {wr[dd.argmax()]:.3e}); at the top-weight samples {np.array2string(dd[top], precision=2)}")
            print("      top weights", np.array2string(wr[top], precision=4), "samples", top, " density", np.array2string(ref[1]["density"][0][top], precision=3))
            if "normals" in ref[1]:
                nn = ref[1]["normals"][0][top]
                print("      |normals| at top", np.array2string(np.linalg.norm(nn, axis=-1), precision=4))


if __name__ == "__main__":
    main()
