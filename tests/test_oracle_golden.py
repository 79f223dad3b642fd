"""Pins oracle/e264_oracle.c to the golden vectors of the reference's own unit tests
(/root/reference/src/edge264_check.c:169-359, lifted into tests/golden/check_vectors.json
by tests/golden/extract_check_vectors.py)."""
import json
import os

import numpy as np
import pytest

from oracle.pyoracle import oracle_chroma_mc, oracle_intra, oracle_luma_mc

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "check_vectors.json")))
STRIDE = 32


def border_buffer():
    """check.c:171-180: buf[32*18], p = buf+80, two rows above and the left column."""
    buf = np.zeros(32 * 20, np.uint8)
    p = 80
    for x in range(-1, 16):
        buf[p + x - 32] = (194 + x * 4) & 255
        buf[p + x - 64] = (198 + x * 4) & 255
    for y in range(16):
        buf[p + y * 32 - 1] = (186 - y * 4) & 255
    return buf, p


@pytest.mark.parametrize("kind,n", [("4x4", 4), ("8x8", 8), ("16x16", 16)])
def test_intra_luma_golden(oracle, kind, n):
    for case in G["intra" + kind]:
        buf, p = border_buffer()
        oracle_intra(oracle, kind, buf, p, STRIDE, case["mode"])
        got = np.lib.stride_tricks.as_strided(buf[p:], (n, n), (STRIDE, 1))
        assert got.flatten().tolist() == case["expect"], case["name"]


def test_intra_chroma_golden(oracle):
    """The reference predicts Cb and Cr in one call over interleaved rows (stride halved,
    intra.c:689-694): with stride 32, even rows are Cb, odd rows Cr.  Our oracle works per
    plane with the true row stride 64."""
    for case in G["intra_chroma"]:
        buf, p = border_buffer()
        big = np.zeros(32 * 40, np.uint8)
        big[:len(buf)] = buf
        # Cb plane = rows p-64, p, p+64 ... ; Cr = rows p-32, p+32 ...
        # left columns: the check buffer only defines p[y*32-1] for y<16 which covers both planes
        oracle_intra(oracle, "chroma", big, p, 64, case["mode"])
        oracle_intra(oracle, "chroma", big, p + 32, 64, case["mode"])
        got = np.lib.stride_tricks.as_strided(big[p:], (16, 8), (32, 1))
        assert got.flatten().tolist() == case["expect"], case["name"]


def inter_src():
    return np.array([((i * 21 + j) * 37) & 255 for i in range(21) for j in range(21)] + [0] * 64, np.uint8)


def test_inter_luma_golden(oracle):
    src = inter_src()
    for idx, case in enumerate(G["inter_luma"]):
        mode = case["mode"]
        w = 4 << (mode >> 4)
        h = 8 if idx < 16 else 16
        xF, yF = mode & 3, mode >> 2 & 3
        # block origin src+44 = row 2, col 2 of the 21x21 patch; frame larger than the patch => no clamping
        got = oracle_luma_mc(oracle, src, 21, 21, 21, 2, 2, xF, yF, w, h)
        assert got.flatten().tolist() == case["expect"], case["name"]


def test_inter_chroma_golden(oracle):
    """decode_inter_chroma handles Cb and Cr rows interleaved (sstride = stride_C/2): the 21-byte
    rows of the patch alternate planes, so one plane has row stride 42 and ABCD={3,21,5,35}
    is xFrac=7,yFrac=5 ((8-7)*(8-5)=3, 7*3=21, 1*5=5, 35)."""
    src = inter_src()
    for case in G["inter_chroma"]:
        cw, rows = case["cols"], case["rows"]
        exp = np.array(case["expect"], np.uint8).reshape(rows, cw)
        for plane in range(2):
            got = oracle_chroma_mc(oracle, src[2 * 21 + 2 + plane * 21:], 42, 64, 64, 0, 0, 7, 5, cw, rows // 2)
            assert np.array_equal(got, exp[plane::2]), (case["w"], case["h"], plane)


def test_golden_vectors_as_packets(oracle):
    """The same vectors as whole-picture command packets (tests/golden_packets.py) through the oracle's FRAME path: pins
    the packet construction that tests/test_hip_golden.py runs on the GPU."""
    from edge264_amd import backend
    from tests import golden_packets as GP
    n = 0
    for name, pkt, init, checks in GP.cases():
        assert backend.packet_check(pkt) == 0, name
        dpb = [init.get(s) for s in range(32)]
        oracle.decode_frame(pkt, dpb, 1)
        GP.check(name, dpb[GP.DST], checks)
        n += 1
    assert n == 14 + 31 + 7 + 7 + 48 + 3
