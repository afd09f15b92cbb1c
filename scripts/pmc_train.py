"""a few training steps in one chain mode (for rocprofv3): python scripts/pmc_train.py <f32|f16x2|bf16>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]] + sys.argv[1:]
exec(open(os.path.join(ROOT, "scripts", "time_train.py")).read())
