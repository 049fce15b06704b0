import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the two-panel (K = 256) driver normally engages only for n >= 4096; tests exercise it on small shapes
os.environ.setdefault("DHQR_PAIR_MIN_N", "512")
# the single-workgroup route for matrices that fit one compute unit (csrc/dhqr_small.h) is the product default; the suite
# keeps exercising the GENERAL drivers on its many small shapes and switches the route on where it tests it
# (Context.set_small_route / DHQR_SMALL=1: test_small_route_* in test_gpu_kernels.py and test_emulated_library.py)
os.environ.setdefault("DHQR_SMALL", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    p = g.import_package()
    if not os.path.exists(p._lib.SO_PATH):  # normally prebuilt in-tree by __graft_entry__.build()
        p.build()
    return p


@pytest.fixture(scope="session")
def orc():
    from oracle import dhqr_oracle
    dhqr_oracle.build()
    return dhqr_oracle


@pytest.fixture(scope="session")
def emulated_so(tmp_path_factory):
    """path of the EMULATED library (tests/simt: csrc/ host-compiled in fiber mode), built once per session"""
    from dist_helpers import CLANG, build_emulated_library
    if not os.path.exists(CLANG):
        pytest.skip("host clang++ (ROCm llvm) not found")
    return build_emulated_library(tmp_path_factory.mktemp("emulated_lib"))
