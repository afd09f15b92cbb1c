"""Per-ray evidence for the float64 gate of the end-to-end RGB parity (round 6).

For every full-size trained-weights case of tests/test_hip_f16x2.py::test_f16x2_full_size_vs_oracle: the HIP path in the
mode of record (f16x2) and in the strict f32 mode, the fp32 CPU oracle (= the reference's arithmetic) and the float64 build of
the SAME oracle (oracle/oracle_f64.py), all on the same rays -> gpurun_out/f64/<case>.npz with the four level-1 (and level-0)
renderings, plus the fine level of both HIP modes fed the fp32 oracle's coarse step function (bin indices, sdist).
Runs on the GPU box:  python scripts/parity_f64_collect.py [case ...]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refnerf_pl_amd  # noqa: E402,F401
from refnerf_pl_amd import _hip as hip, synthetic  # noqa: E402
from oracle import oracle as O, oracle_f64 as O64  # noqa: E402
from helpers import cfg_from_bindings, load_golden, trained_blob, trained_llff_blob, trained_long_blob  # noqa: E402
from test_hip_parity import DEV, dev_rays, run_hip_model  # noqa: E402

F16X2 = 3
CASES = ["C2_trained_like", "C2_trained_long", "C3_trained_long", "C3_trained_long_view2", "C3_trained_long_view3",
         "C4_trained_llff", "C5_trained_llff"]


def case_inputs(case):
    R, N = (8192, 192) if case.startswith("C3") else ((2048, 256) if case.startswith("C5") else (4096, 128))
    kw = {}
    if "trained_long" in case:
        view = {"": 3, "_view2": 11, "_view3": 23}[case.split("trained_long")[1]]
        P, rays = trained_long_blob(), synthetic.blender_rays(R, seed=view, center_frac=0.8)
    elif case.endswith("trained_llff"):
        kw = cfg_from_bindings(load_golden("model_trained_llff_eval")["bindings"])[0]
        P, rays = trained_llff_blob(), synthetic.llff_rays(R, seed=7 if case.startswith("C5") else 3)
    else:
        P, rays = trained_blob(), synthetic.blender_rays(R, seed=3, center_frac=0.8)
    return P, rays, kw, dict(num_prop_samples=N, num_nerf_samples=N), N


def main():
    hip.require_device()
    out_dir = os.path.join(ROOT, "gpurun_out", "f64")
    os.makedirs(out_dir, exist_ok=True)
    for case in (sys.argv[1:] or CASES):
        P, rays, kw, lv, N = case_inputs(case)
        t0 = time.time()
        h16 = run_hip_model(hip, P, rays, kw, lv, precision=F16X2)
        h32 = run_hip_model(hip, P, rays, kw, lv, precision=0)
        t1 = time.time()
        o32 = O.model_forward(P, rays, history=False, **lv, **kw)
        t2 = time.time()
        o64 = O64.model_forward(P, rays, history=False, **lv, **kw)
        t3 = time.time()
        rec = {}
        for L in range(2):
            for tag, res in (("hip16", h16), ("hip32", h32), ("o32", o32), ("o64", o64)):
                rec[f"L{L}_{tag}_rgb"] = res[L]["r_rgb"]
            rec[f"L{L}_idx_hip16_eq_o32"] = (h16[L]["bin_idx"] == o32[L]["bin_idx"]).sum(-1).astype(np.int32)
            rec[f"L{L}_idx_hip32_eq_o32"] = (h32[L]["bin_idx"] == o32[L]["bin_idx"]).sum(-1).astype(np.int32)
            rec[f"L{L}_w_err_hip16_o64"] = np.abs(h16[L]["weights"] - o64[L]["weights"]).max(-1)
            rec[f"L{L}_w_err_o32_o64"] = np.abs(o32[L]["weights"] - o64[L]["weights"]).max(-1)
        # the fine level alone on the fp32 oracle's coarse step function: indices must be bit-identical (shared rn_det_logf / rn_det_expf)
        sub = dev_rays(rays)
        for prec, tag in ((F16X2, "hip16"), (0, "hip32")):
            packed = hip.pack_weights(torch.tensor(P, device=DEV), precision=hip.level_image(prec, False, 0))
            cfg1 = hip.default_cfg(n_samples=N, n_in=N, precision=prec, **kw)
            res = hip.level_forward(packed, cfg1, sub, torch.tensor(o32[0]["sdist"], device=DEV), torch.tensor(o32[0]["weights"], device=DEV))
            rec[f"L1_{tag}_rgb_given_o32_step"] = res["r_rgb"].cpu().numpy()
            rec[f"L1_{tag}_idx_differing_given_o32_step"] = np.int64((res["bin_idx"].cpu().numpy() != o32[1]["bin_idx"]).sum())
            rec[f"L1_{tag}_sdist_bit_equal_given_o32_step"] = np.bool_(np.array_equal(res["sdist"].cpu().numpy(), o32[1]["sdist"]))
        np.savez_compressed(os.path.join(out_dir, case + ".npz"), **rec)
        e16 = np.abs(rec["L1_hip16_rgb"] - rec["L1_o64_rgb"]).max(-1)
        e32 = np.abs(rec["L1_hip32_rgb"] - rec["L1_o64_rgb"]).max(-1)
        er = np.abs(rec["L1_o32_rgb"] - rec["L1_o64_rgb"]).max(-1)
        eo = np.abs(rec["L1_hip16_rgb"] - rec["L1_o32_rgb"]).max(-1)
        print(f"{case}: hip {t1 - t0:.1f}s o32 {t2 - t1:.1f}s o64 {t3 - t2:.1f}s | L1 max |hip16-f64| {e16.max():.3e} |hip32-f64| {e32.max():.3e} "
              f"|o32-f64| {er.max():.3e} |hip16-o32| {eo.max():.3e} | p9999 {np.quantile(e16, .9999):.3e} {np.quantile(e32, .9999):.3e} {np.quantile(er, .9999):.3e} | "
              f"given o32 step: idx differing hip16 {int(rec['L1_hip16_idx_differing_given_o32_step'])} hip32 {int(rec['L1_hip32_idx_differing_given_o32_step'])}, "
              f"sdist bit-equal {bool(rec['L1_hip16_sdist_bit_equal_given_o32_step'])} {bool(rec['L1_hip32_sdist_bit_equal_given_o32_step'])}", flush=True)


if __name__ == "__main__":
    main()
