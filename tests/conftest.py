import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN_REF = os.path.join(ROOT, "tests", "golden", "ref")
REF_CASES = ["randtest1", "randtest2", "randtest3", "rfctest1", "rfctest2", "rfctest3",
             "zerotest1", "zerotest2", "zerotest3"]  # test/Test.hs:56-67 testCases


def read_case(name):
    with open(os.path.join(GOLDEN_REF, name + ".z"), "rb") as f:
        z = f.read()
    with open(os.path.join(GOLDEN_REF, name + ".gold"), "rb") as f:
        g = f.read()
    return z, g


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    """A pzg context on device 0.  Fails (does not skip) when the HIP extension or the device is
    missing: GPU tests must never pass on a fallback."""
    import torch  # first: libpzg.so then binds to the HIP runtime torch has already loaded (as in bench.py); the other
    torch.cuda.init()  # order leaves torch without a device for the tests that keep their arenas in torch tensors
    import pure_zlib_amd as P
    ctx = P.Context(0)
    yield ctx
    ctx.close()
