"""ctypes binding of the float64 "truth" build of the CPU oracle (librefnerf_oracle_f64.so).

TEST INFRASTRUCTURE ONLY (like everything under oracle/): the same restatement as oracle/refnerf_oracle.c, compiled
with the real type switched to double (oracle/refnerf_oracle_f64.h, `make -C oracle librefnerf_oracle_f64.so`).  It is
what the fp32 evaluations of the path -- the reference's own and the HIP kernels' -- are both rounding errors away
from: tests gate `|hip - f64|` by `|fp32 oracle - f64|` on rays whose level-1 sample positions are ill-conditioned
(SURVEY.md H2 does the same for the IDE).  Pinned by tests/test_oracle_golden.py::test_f64_build_*: on every model
fixture the float64 build is within fp32 rounding of the reference's outputs.

Same functions as oracle.oracle (this module executes that file over the other library): float64 arrays in and out.
"""
import os as _os

_REAL_F64 = True
_src = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "oracle.py")
with open(_src) as _fh:
    exec(compile(_fh.read(), _src, "exec"))
