import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def refkernels():
    from oracle.pyoracle import RefKernels
    try:
        return RefKernels()
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref/libe264_refkernels.so not built (needs /root/reference)")


@pytest.fixture(scope="session")
def refdecoder():
    from oracle.pyoracle import ref_decoder
    try:
        return ref_decoder()
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref/libedge264_ref.so not built (needs /root/reference)")
