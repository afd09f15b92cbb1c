import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
import argparse
args = argparse.Namespace(gpus=1, steps=20, warmup=5, config="C2", rays=None, samples=None, precision="f16x2", train_precision="f16x2",
                          no_cpu_baseline=True, no_image=True, no_train=True, no_graph=False, no_other_configs=True, no_rccl=True)
dev = torch.device("cuda", 0)
spec = dict(bench.CONFIGS["C2"])
from refnerf_pl_amd import utils, _hip
model, cfg, blob = bench.build_model(args, spec, dev)
rays = utils.rays_from_dict(bench.make_rays(spec, spec["rays"], seed=1), dev)
for mode in ("f16x2", "f32", "bf16", "f16", "bf16", "f16"):
    cfg.hip_precision = mode
    ts = []
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            model(rays, 1.0, True)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(mode, " ".join("%.2f" % t for t in ts), flush=True)
