#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts/profile_gpu.sh (gpurun_out/prof_<tag>/) into the small
per-round summaries committed under profiles/<round>/:
  kernel_stats.csv        -- rocprofv3 --kernel-trace --stats table of the rn:: kernels
  pmc_<kernel>.csv        -- per-dispatch mean/min/max of every counter collected (separate --pmc passes)
and refresh profiles/traffic.json (HBM-side bytes per launch, gfx950 FETCH_SIZE x2 correction).
usage: summarize_prof.py gpurun_out/prof_r02 profiles/r02 [C2|C3|...]   (the bench.py --config the passes ran with;
entries of configurations other than C2 are keyed "kernel@config" in traffic.json and their files get a _<config> suffix)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "C2"
sfx = "" if cfg == "C2" else "_" + cfg
WORKLOAD = {"C2": "4096 rays x 128 samples, one level per launch", "C3": "8192 rays x 192 samples (shiny network), one level per launch",
            "C4": "4096 LLFF rays x 128 samples, one level per launch", "C5": "16384 (+512 noisy) LLFF rays x 256 samples, one level per launch"}[cfg]
bench_args = "" if cfg == "C2" else f" --config {cfg}"
os.makedirs(dst, exist_ok=True)
KERNELS = {"level_fwd_bf16": "rn::level_fwd_bf16", "level_fwd_f32": "rn::level_fwd_f32",
           "level_fwd_train_f32": "rn::level_fwd_train_f32", "level_bwd_f32": "rn::level_bwd_f32",
           "wgrad_kernel": "rn::wgrad_kernel", "wgrad_bf16x3_kernel": "rn::wgrad_bf16x3_kernel", "level_bwd_bf16c": "rn::level_bwd_bf16c",
           "level_fwd_train_bf16c": "rn::level_fwd_train_bf16c", "level_fwd_f16": "rn::level_fwd_f16",
           "bwd_seed_kernel": "rn::bwd_seed_kernel", "wgrad_reduce": "rn::wgrad_reduce",
           "level_fwd_bf16_ring": "rn::level_fwd_bf16_ring", "level_fwd_f16_ring": "rn::level_fwd_f16_ring",
           "level_fwd_f16x2": "rn::level_fwd_f16x2", "level_fwd_f16x2_ring": "rn::level_fwd_f16x2_ring",
           "level_fwd_train_f16x2c": "rn::level_fwd_train_f16x2c", "level_bwd_f16x2c": "rn::level_bwd_f16x2c",
           "wgrad_f16s_kernel": "rn::wgrad_f16s_kernel", "delta_scale_min": "rn::delta_scale_min",
           "level_fwd_train_sq": "rn::level_fwd_train_sq", "level_fwd_train_sq_h": "rn::level_fwd_train_sq_h", "level_bwd_sq": "rn::level_bwd_sq",
           "wgrad_sq_kernel": "rn::wgrad_sq_kernel", "wgrad_sq256_kernel": "rn::wgrad_sq256_kernel", "delta_kappa_min": "rn::delta_kappa_min",
           "pack_train_chunks": "rn::pack_train_chunks", "pack_train_consts": "rn::pack_train_consts"}

rows = list(csv.reader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
with open(os.path.join(dst, f"kernel_stats{sfx}.csv"), "w", newline="") as f:
    w = csv.writer(f)
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-image --no-other-configs{bench_args}\n")
    w.writerow(rows[0])
    for r in rows[1:]:
        if r and (r[0].startswith("rn::") or r[0].startswith("void rn::")):
            w.writerow(r)

stats = defaultdict(lambda: defaultdict(list))
meta = {}
for path in glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
        for short, full in KERNELS.items():
            if name == full:
                stats[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[short] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
for short, counters in stats.items():
    with open(os.path.join(dst, f"pmc_{short}{sfx}.csv"), "w") as f:
        g = meta[short]
        f.write(f"# rocprofv3 --pmc passes (separate runs) of: python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-image --no-other-configs{bench_args}\n")
        f.write(f"# kernel {KERNELS[short]}, grid {g[0]}, wg {g[1]}, VGPR {g[2]}, AGPR {g[3]}, SGPR {g[4]}, scratch {g[5]}\n")
        f.write("counter,dispatches,mean_per_dispatch,min,max\n")
        for c in sorted(counters):
            v = counters[c]
            f.write(f"{c},{len(v)},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f}\n")

tj = os.path.join(os.path.dirname(dst.rstrip("/")), "traffic.json")
traffic = json.load(open(tj)) if os.path.exists(tj) else {}
for short, counters in stats.items():
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        fs = sum(counters["FETCH_SIZE"]) / len(counters["FETCH_SIZE"])
        ws = sum(counters["WRITE_SIZE"]) / len(counters["WRITE_SIZE"])
        key = KERNELS[short] if cfg == "C2" else f"{KERNELS[short]}@{cfg}"
        e = traffic.get(key, {})
        e.update({"round": os.path.basename(dst.rstrip("/")), "workload": WORKLOAD,
                  "FETCH_SIZE_KB": round(fs, 1), "WRITE_SIZE_KB": round(ws, 1),
                  "bytes_per_launch": int((2 * fs + ws) * 1024)})
        traffic[key] = e
    # matrix-pipe occupancy of the launch: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs,
    # GRBM_GUI_ACTIVE the launch's cycles summed over the 8 XCDs; the sustained clock needs the launch duration (kernel trace)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "GRBM_GUI_ACTIVE" in counters:
        busy = sum(counters["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(counters["SQ_VALU_MFMA_BUSY_CYCLES"])
        gui = sum(counters["GRBM_GUI_ACTIVE"]) / len(counters["GRBM_GUI_ACTIVE"]) / 8.0
        key = KERNELS[short] if cfg == "C2" else f"{KERNELS[short]}@{cfg}"
        e = traffic.get(key, {"round": os.path.basename(dst.rstrip("/")), "workload": WORKLOAD})
        if gui > 0 and busy > 0:
            e["mfma_busy"] = round(busy / (gui * 1024.0), 4)
            e["gui_cycles_per_launch"] = int(gui)
            dur_ns = None
            for r in rows[1:]:
                if r and r[0].replace("void ", "").split("(")[0].split("<")[0] == KERNELS[short]:
                    try:
                        dur_ns = float(r[rows[0].index("AverageNs")])
                    except (ValueError, IndexError):
                        pass
            if dur_ns:
                e["sustained_clock_ghz"] = round(gui / dur_ns, 3)
                e["executed_flop_frac"] = round(e["mfma_busy"] * (gui / dur_ns) / 2.4, 4)      # of the 2.4 GHz peak the roofline is priced at
            if "SQ_INSTS_VALU" in counters and "SQ_INSTS_MFMA" in counters and sum(counters["SQ_INSTS_MFMA"]) > 0:
                e["valu_per_mfma"] = round(sum(counters["SQ_INSTS_VALU"]) / sum(counters["SQ_INSTS_MFMA"]), 2)
            traffic[key] = e
json.dump(traffic, open(tj, "w"), indent=2)
print("wrote", sorted(os.listdir(dst)))
