"""A/B of library builds on the training step: python scripts/ab_train_modes.py lib1.so lib2.so ... (each in its own process;
chain modes via REFNERF_AB_MODES, default "f16x2")"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import refnerf_pl_amd  # noqa: F401
    from refnerf_pl_amd import _hip
    if sys.argv[2] != "-":
        _hip.LIB_PATH = os.path.join(ROOT, sys.argv[2])
    os.environ["REFNERF_BENCH_PROBE"] = "1"
    sys.argv = [sys.argv[0]] + os.environ.get("REFNERF_AB_MODES", "f16x2").split()
    exec(open(os.path.join(ROOT, "scripts", "time_train.py")).read())
else:
    for lib in sys.argv[1:]:
        print("==", lib, flush=True)
        subprocess.call([sys.executable, __file__, "--child", lib])
