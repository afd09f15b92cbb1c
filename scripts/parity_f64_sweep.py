"""More evidence for the float64 gate (round 6): further seeded views of the 2500-step Blender weights and of the LLFF weights,
every ray through the HIP path (f16x2 and f32 modes), the fp32 oracle and its float64 build -> gpurun_out/parity_f64_sweep.json.
The gate of tests/test_hip_f16x2.py::test_f16x2_full_size_vs_oracle is evaluated on every ray; nothing is asserted here.
  python scripts/parity_f64_sweep.py [n_views]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refnerf_pl_amd  # noqa: E402,F401
from refnerf_pl_amd import _hip as hip, synthetic  # noqa: E402
from oracle import oracle as O, oracle_f64 as O64  # noqa: E402
from helpers import cfg_from_bindings, load_golden, trained_llff_blob, trained_long_blob  # noqa: E402
from test_hip_parity import run_hip_model  # noqa: E402

F16X2, TOL = 3, 1e-4
K = {F16X2: 4.0, 0: 2.0}
n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hip.require_device()
cases = [("blender", 8192, 192, s) for s in (31, 47, 59, 71, 83, 97, 101, 113)[:n_views]] + \
        [("blender", 4096, 128, s) for s in (5, 17, 29, 41)[:max(1, n_views // 2)]] + \
        [("llff", 4096, 128, s) for s in (11, 13)] + [("llff", 2048, 256, 19)]
out = {"gate": "|hip - f64| <= max(1e-4, K |fp32 oracle - f64|), K = 4 (f16x2) / 2 (f32 mode)", "cases": []}
tot = {"rays": 0, "viol_f16x2": 0, "viol_f32": 0, "over_f16x2": 0, "over_f32": 0, "over_oracle": 0}
for fam, R, N, seed in cases:
    if fam == "blender":
        P, rays, kw = trained_long_blob(), synthetic.blender_rays(R, seed=seed, center_frac=0.8), {}
    else:
        P, rays = trained_llff_blob(), synthetic.llff_rays(R, seed=seed)
        kw = cfg_from_bindings(load_golden("model_trained_llff_eval")["bindings"])[0]
    lv = dict(num_prop_samples=N, num_nerf_samples=N)
    t0 = time.time()
    h = {p: run_hip_model(hip, P, rays, kw, lv, precision=p) for p in (F16X2, 0)}
    o32 = O.model_forward(P, rays, history=False, **lv, **kw)
    o64 = O64.model_forward(P, rays, history=False, **lv, **kw)
    er = np.abs(o32[1]["r_rgb"] - o64[1]["r_rgb"]).max(-1)
    rec = {"family": fam, "rays": R, "samples": N, "seed": seed, "oracle_f32_vs_f64_max": float(er.max()),
           "oracle_rays_over_1e-4": int((er > TOL).sum()), "seconds": round(time.time() - t0, 1)}
    for p, tag in ((F16X2, "f16x2"), (0, "f32")):
        e = np.abs(h[p][1]["r_rgb"] - o64[1]["r_rgb"]).max(-1)
        eo = np.abs(h[p][1]["r_rgb"] - o32[1]["r_rgb"]).max(-1)
        rec[tag + "_vs_f64_max"] = float(e.max())
        rec[tag + "_vs_oracle_f32_max"] = float(eo.max())
        rec[tag + "_vs_oracle_f32_p9999"] = float(np.quantile(eo, 0.9999))
        rec[tag + "_rays_over_1e-4_vs_f64"] = int((e > TOL).sum())
        rec[tag + "_gate_violations"] = int((e > np.maximum(TOL, K[p] * er)).sum())
        over = e > TOL
        rec[tag + "_worst_ratio_on_rays_over"] = float((e[over] / np.maximum(er[over], 1e-30)).max()) if over.any() else 0.0
        rec[tag + "_bin_idx_differing"] = int((h[p][1]["bin_idx"] != o32[1]["bin_idx"]).sum())
        tot["viol_" + tag] += rec[tag + "_gate_violations"]
        tot["over_" + tag] += rec[tag + "_rays_over_1e-4_vs_f64"]
    tot["rays"] += R
    tot["over_oracle"] += rec["oracle_rays_over_1e-4"]
    out["cases"].append(rec)
    print(json.dumps(rec), flush=True)
out["total"] = tot
print("TOTAL", json.dumps(tot))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_f64_sweep.json"), "w"), indent=1)
