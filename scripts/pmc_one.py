"""one level launch pair of a precision mode (for rocprofv3 --pmc): python scripts/pmc_one.py <lib.so | -> <precision int>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import refnerf_pl_amd
from refnerf_pl_amd import _hip, synthetic
if sys.argv[1] != "-":
    _hip.LIB_PATH = os.path.join(ROOT, sys.argv[1])
from test_hip_parity import run_hip_model
P = synthetic.make_params(0, 0.05, 20.0)
rays = synthetic.blender_rays(4096, seed=1, center_frac=0.5)
for _ in range(3):
    run_hip_model(_hip, P, rays, {}, {}, precision=int(sys.argv[2]))
torch.cuda.synchronize()
