"""The reference's own unit-test vectors (src/edge264_check.c:169-359, lifted into tests/golden/check_vectors.json by
tests/golden/extract_check_vectors.py) as COMMAND PACKETS: every vector becomes one 48 x 48 picture whose macroblock
(1, 1) runs the predictor under test on the vector's neighbour / reference samples, so that a whole-frame decoder (the
HIP kernels through the C-ABI, or the oracle's frame path) can be held against data the reference itself publishes --
no other implementation in between.

cases() yields (name, packet, initial DPB contents, (plane, y0, x0, expected 2-D array) checks)."""
import json
import os

import numpy as np

from edge264_amd import packet as P, synth

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "check_vectors.json")))
W = H = 3                      # macroblocks
DST, REF = 0, 1                # DPB slots


def _planes(buf):
    return P.split_planes(buf, W, H)


def _blank(value=0):
    return np.full(P.frame_bytes(W, H) + 64, value, np.uint8)


def _slice(b, ftype="I"):
    return b.add_slice(slice_type={"I": 2, "P": 0}[ftype], first_mb=0, disable_deblocking_filter_idc=1)


def _intra_border(buf):
    """check.c:171-180 around macroblock (1, 1): the row above (and the one above that: unused by 8-bit prediction), the
    corner and the left column.  p[x - stride] = 194 + 4x for x = -1..15, p[y * stride - 1] = 186 - 4y."""
    Y, Cb, Cr = _planes(buf)
    for x in range(-1, 16):
        Y[15, 16 + x] = (194 + 4 * x) & 255
    for y in range(16):
        Y[16 + y, 15] = (186 - 4 * y) & 255
    # chroma: the reference predicts Cb and Cr in one call over interleaved rows of HALF the stride (intra.c:689-694), so
    # in the check buffer even rows are Cb, odd rows Cr: Cb's row above is p - 2 * stride (198 + 4x), Cr's is p - stride
    # (194 + 4x); the left column alternates (Cb 186 - 8j, Cr 182 - 8j)
    for x in range(-1, 8):
        Cb[7, 8 + x] = (198 + 4 * x) & 255
        Cr[7, 8 + x] = (194 + 4 * x) & 255
    for j in range(8):
        Cb[8 + j, 7] = (186 - 8 * j) & 255
        Cr[8 + j, 7] = (182 - 8 * j) & 255


def _intra_packet(kind, **kw):
    b = P.PacketBuilder(W, H, DST, 0)
    s = _slice(b)
    kw.setdefault("flags", 0)
    b.set_mb(4, kind=kind, slice_idx=s, qp=(26, 26, 26), **kw)
    return b.finish()


def intra_cases():
    for case in G["intra4x4"]:
        modes = [synth.I4_DC_AB] * 16            # the other 15 blocks predict 128 from nothing
        modes[0] = case["mode"]
        dpb = _blank()
        _intra_border(dpb)
        pkt = _intra_packet(P.MB_I4x4, modes=modes, chroma_mode=synth.IC_DC_AB)
        yield case["name"], pkt, {DST: dpb}, [(0, 16, 16, np.array(case["expect"], np.uint8).reshape(4, 4))]
    for case in G["intra8x8"]:
        modes = [case["mode"], synth.I8_DC_AB, synth.I8_DC_AB, synth.I8_DC_AB]
        dpb = _blank()
        _intra_border(dpb)
        pkt = _intra_packet(P.MB_I8x8, modes=modes, chroma_mode=synth.IC_DC_AB, flags=P.MBF_T8x8)
        yield case["name"], pkt, {DST: dpb}, [(0, 16, 16, np.array(case["expect"], np.uint8).reshape(8, 8))]
    for case in G["intra16x16"]:
        dpb = _blank()
        _intra_border(dpb)
        pkt = _intra_packet(P.MB_I16x16, i16_mode=case["mode"], chroma_mode=synth.IC_DC_AB)
        yield case["name"], pkt, {DST: dpb}, [(0, 16, 16, np.array(case["expect"], np.uint8).reshape(16, 16))]
    for case in G["intra_chroma"]:
        dpb = _blank()
        _intra_border(dpb)
        pkt = _intra_packet(P.MB_I16x16, i16_mode=synth.I16_DC_AB, chroma_mode=case["mode"])
        exp = np.array(case["expect"], np.uint8).reshape(16, 8)   # rows alternate Cb, Cr
        yield case["name"], pkt, {DST: dpb}, [(1, 8, 8, exp[0::2]), (2, 8, 8, exp[1::2])]


def _inter_src():
    return np.array([((i * 21 + j) * 37) & 255 for i in range(21) for j in range(21)] + [0] * 64, np.uint8)


def _inter_packet(mvs_of_block):
    """P picture, macroblock (1, 1) predicted from slot REF with one vector per 4x4 block (zig order)."""
    b = P.PacketBuilder(W, H, DST, 1)
    s = _slice(b, "P")
    refPic = np.full(8, -1, np.int8)
    refIdx = np.full(8, -1, np.int8)
    refPic[:4] = REF
    refIdx[:4] = 0
    mvs = np.zeros((2, 16, 2), np.int16)
    mvs[0] = mvs_of_block
    b.set_mb(4, kind=P.MB_INTER, slice_idx=s, qp=(26, 26, 26), flags=0, motion=dict(refPic=refPic, refIdx=refIdx, mvs=mvs))
    return b.finish()


def _block_mvs(w, h, mv):
    """The w x h partition at the macroblock's origin moves by mv, every other 4x4 block by (0, 0)."""
    out = np.zeros((16, 2), np.int16)
    for k in range(16):
        if int(P.BX[k]) < w and int(P.BY[k]) < h:
            out[k] = mv
    return out


def inter_cases():
    src = _inter_src()
    patch = src[:441].reshape(21, 21)
    for idx, case in enumerate(G["inter_luma"]):
        mode = case["mode"]
        w, h = 4 << (mode >> 4), (8 if idx < 16 else 16)
        xF, yF = mode & 3, mode >> 2 & 3
        ref = _blank()
        Y, _, _ = _planes(ref)
        Y[14:35, 14:35] = patch                  # the block's origin src + 44 = row 2, column 2 of the patch (check.c:344)
        pkt = _inter_packet(_block_mvs(w, h, (xF, yF)))
        yield case["name"], pkt, {DST: _blank(), REF: ref}, [(0, 16, 16, np.array(case["expect"], np.uint8).reshape(h, w))]
    for case in G["inter_chroma"]:
        # check.c:350-357: Cb and Cr rows interleaved in the 21-byte-row patch (one plane = every other row, row stride 42),
        # origin src + 44, ABCD = {3, 21, 5, 35} = xFrac 7, yFrac 5 in eighths: the luma vector (7, 5)
        cw, rows = case["cols"], case["rows"]
        ref = _blank()
        _, Cb, Cr = _planes(ref)
        for pl, plane in enumerate((Cb, Cr)):
            for j in range(rows // 2 + 1):
                plane[8 + j, 8:8 + cw + 1] = src[44 + pl * 21 + j * 42:44 + pl * 21 + j * 42 + cw + 1]
        pkt = _inter_packet(_block_mvs(case["w"], case["h"], (7, 5)))
        exp = np.array(case["expect"], np.uint8).reshape(rows, cw)
        yield f"INTER_CHROMA_{case['w']}x{case['h']}", pkt, {DST: _blank(), REF: ref}, [(1, 8, 8, exp[0::2]), (2, 8, 8, exp[1::2])]


def cases():
    yield from intra_cases()
    yield from inter_cases()


def check(name, frame, checks):
    planes = _planes(frame)
    for pl, y0, x0, exp in checks:
        got = planes[pl][y0:y0 + exp.shape[0], x0:x0 + exp.shape[1]]
        assert np.array_equal(got, exp), f"{name}: plane {pl}\n got {got.tolist()}\n want {exp.tolist()}"
