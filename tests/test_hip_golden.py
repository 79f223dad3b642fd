"""The reference's published vectors (src/edge264_check.c:185-357) straight through the HIP kernels: every vector is a
whole-picture command packet (tests/golden_packets.py) submitted through the C-ABI; the samples that come back are compared
with the numbers the reference's own unit test holds -- no oracle in between."""
import numpy as np
import pytest

from edge264_amd import backend, packet as P
from tests import golden_packets as GP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    dev = backend.Device(0)  # raises if the extension or the GPU is missing: no silent fallback
    yield dev
    dev.close()


def test_check_vectors_through_the_kernels(device):
    st = backend.Stream(device, GP.W, GP.H)
    nb = P.frame_bytes(GP.W, GP.H)
    try:
        for s in (GP.DST, GP.REF):
            st.alloc(s)
        n = 0
        for name, pkt, init, checks in GP.cases():
            for s, buf in init.items():
                st.upload(s, buf[:nb])
            st.submit(pkt)
            GP.check(name, st.download(GP.DST), checks)
            n += 1
        assert n == 110
    finally:
        st.close()
