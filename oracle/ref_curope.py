"""TEST INFRASTRUCTURE ONLY.  Loader for oracle/_ref/curope.so = the reference's own CPU RoPE-2D
(/root/reference/src/croco/models/curope/curope.cpp:11-65) compiled by oracle/Makefile.

The CUDA half of the reference op (kernels.cu) cannot be built here (no nvcc), so its symbol
`rope_2d_cuda` is left undefined in the .so; we load with RTLD_LAZY so that the CPU branch
(`tokens.is_cuda() == false`) works and the undefined symbol is never bound.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import importlib.machinery
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "curope.so")
_mod = None


def available() -> bool:
    return os.path.isfile(_SO)


def load():
    """Return the reference pybind module exposing rope_2d(tokens[B,N,H,D] f32, pos[B,N,2] i64, base, fwd)."""
    global _mod
    if _mod is not None:
        return _mod
    if not available():
        raise FileNotFoundError(f"{_SO} missing: run `make -C oracle ref` where /root/reference exists")
    import torch  # noqa: F401  (libtorch must be loaded first)

    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        loader = importlib.machinery.ExtensionFileLoader("curope", _SO)
        spec = importlib.util.spec_from_file_location("curope", _SO, loader=loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    _mod = mod
    return mod


def rope_2d_ref(tokens, positions, base: float, fwd: float):
    """In-place reference RoPE on a float32 CPU tensor view (B,N,H,D)."""
    load().rope_2d(tokens, positions, float(base), float(fwd))
    return tokens
