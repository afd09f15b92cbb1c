"""Import shim for the upstream reference (build container only).

TEST INFRASTRUCTURE. Used by ``make_golden.py`` (and nothing else) to import
``/root/reference/internal/*`` on a box that lacks gin/absl/lightning/cv2/...
The stubs are synthesised into a temporary directory at run time; nothing is
written into the reference tree and no reference source is copied here.

The reference never travels to the GPU box: only the ``.npz`` vectors that
``make_golden.py`` emits do.
"""
import math
import os
import sys
import tempfile
import textwrap

REFERENCE_ROOT = os.environ.get("REFNERF_REFERENCE_ROOT", "/root/reference")

_GIN_STUB = '''
import contextlib, ast
_BINDINGS = {}
_REGISTRY = {}

def configurable(arg=None, **_kw):
    def wrap(cls_or_fn):
        name = cls_or_fn.__name__
        _REGISTRY[name] = cls_or_fn
        if isinstance(cls_or_fn, type):
            orig_init = cls_or_fn.__init__
            def __init__(self, *a, **k):
                merged = dict(_BINDINGS.get(name, {}))
                merged.update(k)
                orig_init(self, *a, **merged)
            cls_or_fn.__init__ = __init__
            return cls_or_fn
        def fn(*a, **k):
            merged = dict(_BINDINGS.get(name, {})); merged.update(k)
            return cls_or_fn(*a, **merged)
        return fn
    if callable(arg) and not isinstance(arg, str):
        return wrap(arg)
    return wrap

def bind(name, param, value):
    _BINDINGS.setdefault(name, {})[param] = value

def clear_config():
    _BINDINGS.clear()

def add_config_file_search_path(_p):
    pass

def parse_config_files_and_bindings(files, bindings, skip_unknown=True):
    lines = []
    for f in (files or []):
        with open(f) as fh:
            lines += fh.read().replace('\\\\\\n', ' ').splitlines()
    lines += list(bindings or [])
    for ln in lines:
        ln = ln.split('#', 1)[0].strip() if "'" not in ln else ln.strip()
        if not ln or ln.startswith('#') or '=' not in ln:
            continue
        lhs, rhs = ln.split('=', 1)
        name, param = lhs.strip().rsplit('.', 1)
        rhs = rhs.strip()
        try:
            val = ast.literal_eval(rhs)
        except Exception:
            val = rhs
        bind(name, param, val)

def config_str():
    return repr(_BINDINGS)

@contextlib.contextmanager
def config_scope(_name):
    yield
'''


def _write(root, rel, body):
    path = os.path.join(root, rel)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        fh.write(textwrap.dedent(body))


def install():
    """Put stub modules + the reference on sys.path; returns the stub dir."""
    import numpy as np
    if not hasattr(np, "math"):
        np.math = math  # ref_utils.py:55 uses np.math (removed in NumPy 2)
    stub = tempfile.mkdtemp(prefix="refnerf_stubs_")
    _write(stub, "gin/__init__.py", _GIN_STUB)
    _write(stub, "gin/torch.py", "")
    _write(stub, "absl/__init__.py", "")
    _write(stub, "absl/flags.py", """
        class _F:
            gin_configs = None
            gin_bindings = None
        FLAGS = _F()
        def DEFINE_string(*a, **k): pass
        def DEFINE_multi_string(*a, **k): pass
    """)
    _write(stub, "dm_pix.py", "def ssim(*a, **k):\n    raise NotImplementedError\n")
    _write(stub, "lpips.py", "class LPIPS:\n    def __init__(self, *a, **k):\n        pass\n")
    _write(stub, "cv2.py", "")
    _write(stub, "pycolmap.py", "class SceneManager:\n    pass\n")
    _write(stub, "flatdict.py", "class FlatDict(dict):\n    pass\n")
    _write(stub, "mediapy.py", "")
    _write(stub, "pytorch_lightning/__init__.py",
           "class LightningModule:\n    pass\n")
    sys.path.insert(0, stub)
    sys.path.insert(0, REFERENCE_ROOT)
    import torch
    torch.cuda.synchronize = lambda *a, **k: None  # models.py:783
    return stub


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "internal"))
