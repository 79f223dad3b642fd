"""Conformance clips, when somebody provides them (the 109 JVT clips of the reference's own test are not
in the repository and there is no network).  Same directory convention as the reference's `edge264_test`
(src/edge264_test.c:276-286): `X.264` + `X.yuv` (+ `X.1.yuv`, the second view of MVC clips), in `conformance/` at the
repo root or in $E264_CONFORMANCE_DIR.  Without clips everything here is skipped.

CPU: reference front end + our emitters + oracle replay (capture sink) against X.yuv.
GPU (-m gpu): the same through libedge264_hip.so (HIP sink).
A clip the reference's parser does not support (ENOTSUP: fields, MBAFF, 4:2:2, >8 bit ...) is skipped, like the reference's
own test reports it as unsupported instead of failing.
"""
import errno
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DIR = os.environ.get("E264_CONFORMANCE_DIR", os.path.join(ROOT, "conformance"))
CLIPS = sorted(glob.glob(os.path.join(DIR, "*.264")))
FRONT = os.path.join(ROOT, "edge264_amd", "libedge264_hipfront.so")

# (the clip tests parametrise over CLIPS: none collected when the directory is empty; tests/test_conformance_harness.py proves on
# generated clips, laid out the same way, that the code below is alive)


def check(clip, frames, codes):
    if errno.ENOTSUP in codes:
        pytest.skip("unsupported by the reference's parser (ENOTSUP)")
    base = clip[:-4]
    views = [base + ".yuv"] + ([base + ".1.yuv"] if os.path.exists(base + ".1.yuv") else [])
    for v, path in enumerate(views):
        want = np.fromfile(path, np.uint8)
        got = np.concatenate([p.ravel() for fr in frames for p in fr[3 * v:3 * v + 3]]) if frames else np.zeros(0, np.uint8)
        assert got.size == want.size, f"{os.path.basename(path)}: {got.size} bytes decoded, {want.size} expected"
        if not np.array_equal(got, want):
            first = int(np.nonzero(got != want)[0][0])
            per = sum(p.size for p in frames[0][3 * v:3 * v + 3])
            pytest.fail(f"{os.path.basename(path)}: first difference in frame {first // per} at byte {first % per}")


def run_capture(clip, oracle):
    """one clip through reference front end + emitters + oracle replay (CPU)"""
    from oracle.pyoracle import HipFront
    frames, codes, _ = HipFront().decode_capture(open(clip, "rb").read(), oracle)
    check(clip, frames, codes)


def run_hip(clip):
    """one clip through reference front end + emitters + libedge264_hip.so (GPU)"""
    from oracle.pyoracle import HipFront
    h = HipFront()
    h.lib.e264front_set_sink(0)
    frames, codes = h.decode(open(clip, "rb").read())
    check(clip, frames, codes)


@pytest.mark.parametrize("clip", CLIPS, ids=[os.path.basename(c) for c in CLIPS])
def test_clip_capture_oracle(clip, oracle):
    run_capture(clip, oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("clip", CLIPS, ids=[os.path.basename(c) for c in CLIPS])
def test_clip_hip(clip):
    run_hip(clip)
