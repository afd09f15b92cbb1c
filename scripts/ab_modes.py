"""A/B of library builds on the eval forward: python scripts/ab_modes.py rays samples lib1.so lib2.so ... (each build in its own
child process, two rounds; modes f16x2, bf16, f16)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import refnerf_pl_amd  # noqa: F401
    from refnerf_pl_amd import _hip
    if sys.argv[2] != "-":
        _hip.LIB_PATH = os.path.join(ROOT, sys.argv[2])
    sys.argv = [sys.argv[0]] + sys.argv[3:]
    exec(open(os.path.join(ROOT, "scripts", "time_modes.py")).read())
else:
    R, N = sys.argv[1], sys.argv[2]
    for rep in range(2):
        for lib in sys.argv[3:]:
            print("==", lib, flush=True)
            subprocess.call([sys.executable, __file__, "--child", lib, R, N, "f16x2", "bf16", "f16"])
