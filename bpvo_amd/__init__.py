"""bpvo_amd — MI355X-native dense photometric alignment (bit-planes visual odometry hot path).

Host-side Python here is a thin ctypes mirror of the C ABI in include/bpvo_hip/c_api.h (used by the tests and
bench.py); the product is csrc/ (hand-written HIP kernels + the C-ABI shared library libbpvo_hip.so) and the C++
facade include/bpvo_hip/vo.hpp.  There is no CPU fallback: `load()` raises if the HIP library is not built.
"""
import os

from . import capi, synth  # noqa: F401
from .capi import Binding, BpvoError, Params  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbpvo_hip.so")


def load() -> Binding:
    """Load libbpvo_hip.so (built by __graft_entry__.build()). Fails loudly when it is missing."""
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    return Binding(LIB_PATH, "bpvo_hip_")
