"""The reference's OWN test programs, unchanged, on the product (the acid test of "drops in").

oracle/Makefile `programs` compiles /root/reference/src/edge264_test.c and src/edge264_check.c as they are and links them against
edge264_amd/libedge264_hipfront.so (the seven edge264.h functions over the HIP back end) where the reference links its own libedge264:
  * edge264_check_hip   the NAL / API return-code cases of src/edge264_check.c:438-444 (supp-nals, unsupp-nals, max-logs, finish-frame,
                        nal-ref-idc-0, page-boundaries: every edge264_decode_NAL answer and the frame count compared by the program itself)
  * edge264_test_hip    the conformance / benchmark runner of src/edge264_test.c:207-272, 427-546: decodes X.264, compares EVERY MACROBLOCK of
                        every frame with X.yuv itself, `-b` prints time: / CPU: / memory:
The binaries are test-side (oracle/_ref/, git-ignored, they travel with the snapshot).  X.yuv is written here, at test time, by the UNMODIFIED
reference decoder (oracle/_ref/libedge264_ref.so) -- 93 MB per 1080p fixture is too much to commit; nothing on the GPU box reads /root/reference."""
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.path.join(ROOT, "oracle", "_ref")
STREAMS = os.path.join(HERE, "golden", "streams")
API = os.path.join(HERE, "golden", "api")
ANSI = re.compile(r"\x1b\[[0-9;]*[A-Za-z]")

pytestmark = pytest.mark.gpu


def program(name):
    p = os.path.join(REF, name)
    if not os.path.exists(p):
        pytest.fail(f"{p} missing: built in the container by `make -C oracle programs` (__graft_entry__.build()), travels with the snapshot")
    return p


def test_edge264_check_api_cases(tmp_path):
    """src/edge264_check.c run from a directory whose tests/ holds the reference's fixtures: it exits 0 and counts 10 PASS -- 7 API cases
    (its own ASSERTs on every NAL's return code), page-boundaries, and the two kernel-vector suites it carries in itself."""
    os.symlink(API, tmp_path / "tests")
    r = subprocess.run([program("edge264_check_hip")], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    out = ANSI.sub("", r.stdout)
    assert r.returncode == 0, out[-2000:] + r.stderr[-2000:]
    last = [l for l in out.splitlines() if l.strip()][-1]
    assert re.fullmatch(r"10 PASS", last.strip()), out[-1500:]


@pytest.mark.parametrize("name", ["hd1080_ipp30", "cabac_hd1080_ibbp30", "nat1080_ipp30", "cabac_nat1080_ibbp30"])
def test_edge264_test_benchmark_mode(name, tmp_path):
    """`edge264_test -b X.264` with X.yuv beside it: the program's own per-macroblock compare says PASS (1 PASS, 0 UNSUPPORTED, 0 FAIL) and it
    prints its time: / CPU: / memory: lines -- the figure the reference's README quotes for itself, here for ONE stream through the GPU path."""
    import numpy as np
    from oracle.pyoracle import ref_decoder
    data = open(os.path.join(STREAMS, name + ".264"), "rb").read()
    frames, _ = ref_decoder().decode(data)  # the unmodified reference on the host: cropped planes in output order
    assert len(frames) >= 12
    with open(tmp_path / (name + ".yuv"), "wb") as f:
        for fr in frames:
            for plane in fr:
                f.write(np.ascontiguousarray(plane).tobytes())
    os.symlink(os.path.join(STREAMS, name + ".264"), tmp_path / (name + ".264"))
    r = subprocess.run([program("edge264_test_hip"), name + ".264", "-b"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    out = ANSI.sub("", r.stdout)
    assert r.returncode == 0, out[-2000:] + r.stderr[-2000:]
    assert "1 PASS, 0 UNSUPPORTED, 0 FAIL" in out, out[-1500:]
    assert "Erroneous macroblock" not in out
    m = re.search(r"time: ([0-9.]+)s\nCPU: ([0-9.]+)s\nmemory: ([0-9.]+)MB", out)
    assert m, out[-500:]
    print(f"{name}: edge264_test -b on the GPU library: {len(frames)} frames, time {m.group(1)} s, CPU {m.group(2)} s, memory {m.group(3)} MB")
