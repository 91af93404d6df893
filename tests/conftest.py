import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """ctypes binding of the CPU oracle (test infrastructure). Built on demand with gcc."""
    import __graft_entry__ as ge
    from bpvo_amd import capi
    if not os.path.exists(ge.ORACLE_LIB):
        ge.build_oracle()
    return capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")


@pytest.fixture(scope="session")
def hip():
    """ctypes binding of the product library; the tests that use it are marked gpu."""
    # Some GPU tests also use torch for device buffers.  This image's torch wheel carries its own copy of the HIP / HSA runtime, and
    # that copy finds no GPU when it is initialised AFTER the system runtime the product library links (the other order works): bring
    # torch's up first, so that the outcome does not depend on which tests were selected.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    import bpvo_amd
    return bpvo_amd.load()
