"""REFNERF_PROF=1 stamps + per-wave DMA-wait / barrier-wait cycles of the split-f16 eval kernel from a -DREFNERF_PROF_WAITS build:
  python scripts/prof_split_waits.py ab/pw.so"""
import os, sys
os.environ["REFNERF_PROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import refnerf_pl_amd  # noqa
from refnerf_pl_amd import _hip
_hip.LIB_PATH = os.path.join(ROOT, sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
exec(open(os.path.join(ROOT, "scripts", "prof_split_phases.py")).read())
