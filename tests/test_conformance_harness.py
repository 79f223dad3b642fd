"""The conformance harness checked against itself: the day the JVT clips arrive it must not be the harness
that fails.  Three generated fixtures -- CAVLC I/P/B, CABAC with 8x8 transform and scaling lists, and an MVC stream with two
views -- are laid out exactly as the reference's own test expects its corpus (/root/reference/src/edge264_test.c:276-286:
`X.264` + `X.yuv`, and `X.1.yuv` for the second view), the .yuv files written from what the UNMODIFIED reference decoder
(oracle/_ref/libedge264_ref.so) outputs, and tests/test_conformance.py's own code is run on that directory: through the capture sink
and the oracle on the CPU, through the HIP sink on the GPU.  A fourth "clip" with one flipped sample must FAIL, so that a check
that compares nothing cannot pass."""
import os
import shutil

import numpy as np
import pytest

import test_conformance as tc

HERE = os.path.dirname(os.path.abspath(__file__))
STREAMS = os.path.join(HERE, "golden", "streams")
REFLIB = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libedge264_ref.so")
NAMES = ["ipb_spatial", "cabac_t8x8_scaling", "mvc_ipb"]

pytestmark = pytest.mark.skipif(not os.path.exists(REFLIB) or not os.path.exists(tc.FRONT), reason="reference decoder / front end not built (need /root/reference once)")


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    from oracle.pyoracle import ref_decoder
    d = tmp_path_factory.mktemp("conformance")
    ref = ref_decoder()
    for name in NAMES:
        src = os.path.join(STREAMS, name + ".264")
        shutil.copy(src, d / (name + ".264"))
        frames, codes = ref.decode(open(src, "rb").read())
        assert frames, name
        views = len(frames[0]) // 3
        for v in range(views):
            with open(d / (name + (".yuv" if v == 0 else ".1.yuv")), "wb") as f:
                for fr in frames:
                    for p in fr[3 * v:3 * v + 3]:
                        f.write(np.ascontiguousarray(p).tobytes())
    # a clip whose expected output is wrong in ONE sample: the harness must notice
    shutil.copy(d / "ipb_spatial.264", d / "broken.264")
    want = bytearray(open(d / "ipb_spatial.yuv", "rb").read())
    want[len(want) // 2] ^= 1
    open(d / "broken.yuv", "wb").write(bytes(want))
    return d


def test_layout_has_a_second_view(corpus):
    assert os.path.exists(corpus / "mvc_ipb.1.yuv") and not os.path.exists(corpus / "ipb_spatial.1.yuv")


@pytest.mark.parametrize("name", NAMES)
def test_harness_on_generated_clips_cpu(corpus, name, oracle):
    tc.run_capture(str(corpus / (name + ".264")), oracle)


def test_harness_notices_a_wrong_sample_cpu(corpus, oracle):
    with pytest.raises(pytest.fail.Exception, match="first difference"):
        tc.run_capture(str(corpus / "broken.264"), oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_harness_on_generated_clips_gpu(corpus, name):
    tc.run_hip(str(corpus / (name + ".264")))


@pytest.mark.gpu
def test_harness_notices_a_wrong_sample_gpu(corpus):
    with pytest.raises(pytest.fail.Exception, match="first difference"):
        tc.run_hip(str(corpus / "broken.264"))
