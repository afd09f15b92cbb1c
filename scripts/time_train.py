"""training step timing per chain mode at C2 shape: python scripts/time_train.py [modes...]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
args = argparse.Namespace(gpus=1, steps=12, warmup=3, config="C2", rays=None, samples=None, precision="f16x2", train_precision="f32",
                          no_cpu_baseline=True, no_image=True, no_train=False, no_graph=False, no_other_configs=True)
dev = torch.device("cuda", 0)
spec = dict(bench.CONFIGS["C2"])
if len(sys.argv) > 1 and sys.argv[1].isdigit():
    spec["rays"], spec["samples"] = int(sys.argv[1]), int(sys.argv[2]); sys.argv = sys.argv[:1] + sys.argv[3:]
from refnerf_pl_amd import utils
model, cfg, blob = bench.build_model(args, spec, dev)
if os.environ.get("REFNERF_WGRAD_MODE"):
    cfg.hip_wgrad_mode = os.environ["REFNERF_WGRAD_MODE"]
if os.environ.get("REFNERF_NO_FINITE_CHECK"):
    cfg.hip_check_finite = False          # timing-only builds (-DREFNERF_EXPERIMENT_*) compute garbage
rays = utils.rays_from_dict(bench.make_rays(spec, spec["rays"], seed=1), dev)
sync = torch.cuda.synchronize
for mode in (sys.argv[1:] or ["f32", "f16x2", "bf16"]):
    r = bench.train_step_bench(args, spec, model, cfg, rays, 0, 1, dev, None, sync, lambda x: x, mode, n_steps=6, n_warm=2)
    print(mode, "%.2f ms/step" % r["ms_per_step"], "loss %.6f" % r["loss"], {k: (round(v["avg_launch_ms"], 3), round(v["frac"], 3)) for k, v in r["kernels"].items()}, flush=True)
