"""e264_dbkparam2_kernel's source run on the HOST (tests/emu): bS against the oracle's derivation (edge264_deblock.c:958-1118),
alpha / beta / indexA against a direct restatement of edge264_deblock.c:945-955, on frames with every macroblock kind,
multiple slices, all deblocking modes and filter offsets.  The -m gpu tests cover the same kernel through whole-frame parity."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from edge264_amd import packet as P, synth
from oracle.pyoracle import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
ALPHA = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255]
BETA = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18]


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(HERE, "emu")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(d, "libe264_pred_emu.so"))
    lib.e264emu_dbkparam_raw.argtypes = [C.c_char_p, C.c_void_p]
    lib.e264emu_dbkparam_raw.restype = C.c_int
    lib.e264emu_dbkparam_frame2.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
    lib.e264emu_dbkparam_frame2.restype = C.c_int
    lib.e264emu_dbkparam_frame2_nol1.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
    lib.e264emu_dbkparam_frame2_nol1.restype = C.c_int
    lib.e264emu_dbk_pieces.argtypes = [C.c_void_p, C.c_void_p]
    lib.e264emu_dbk_pieces.restype = None
    return lib


def expected_ab(pk, w):
    """bytes 32..63 of every record: alpha[9], beta[9], indexA[9], zero tail"""
    n = len(pk.mbs)
    out = np.zeros((n, 32), np.uint8)
    for a in range(n):
        m = pk.mbs[a]
        if not (m["flags"] & P.MBF_DEBLOCK) or m["kind"] == P.MB_ABSENT:
            continue
        s = pk.slices[int(m["slice"])]
        for pl in range(3):
            for t in range(3):
                qm = int(m["qp"][pl])
                qn = qm
                if t == 1 and m["flags"] & P.MBF_EDGE_LEFT:
                    qn = int(pk.mbs[a - 1]["qp"][pl])
                if t == 2 and m["flags"] & P.MBF_EDGE_TOP:
                    qn = int(pk.mbs[a - w]["qp"][pl])
                qpav = (qm + qn + 1) >> 1
                ia = min(max(qpav + int(s["FilterOffsetA"]), 0), 51)
                ib = min(max(qpav + int(s["FilterOffsetB"]), 0), 51)
                out[a, pl * 3 + t] = ALPHA[ia]
                out[a, 9 + pl * 3 + t] = BETA[ib]
                out[a, 18 + pl * 3 + t] = ia
    return out


CASES = [
    ("ipb", "IPB", 9, 7, dict()),
    ("wide_rows", "IPB", 70, 3, dict(num_refs=2)),          # a workgroup's 64 macroblocks span row ends
    ("slices_idc2", "IPBP", 11, 6, dict(slices_per_frame=4, deblock_idc=2, filter_offsets=(6, -4))),
    ("t8x8_pcm", "IPB", 7, 5, dict(t8x8=True, pcm_prob=0.2, i_kinds=(P.MB_I4x4, P.MB_I8x8, P.MB_I16x16))),
    ("no_deblock", "IP", 5, 4, dict(deblock=False)),
    ("tiny", "IPB", 1, 1, dict()),
    ("extreme_vectors", "IPBB", 8, 5, dict(mv_range=2047, num_refs=2, p_skip=0.0)),  # differences beyond int16: the packed compare must not wrap
    ("small_differences", "IPBB", 9, 6, dict(mv_range=3, num_refs=1, residual_prob=0.05)),  # |dx|, |dy| around the threshold of 4
]


@pytest.mark.parametrize("name,gop,w,h,kw", CASES, ids=[c[0] for c in CASES])
def test_dbkparam_emu(emu, name, gop, w, h, kw):
    orc = Oracle()
    g = synth.StreamSynth(w, h, seed=len(name) * 13, **kw)
    for ft in gop:
        pkt = g.next_frame(ft)
        pk = P.Packet(pkt)
        n = w * h
        got = np.full((n, 64), 0x5A, np.uint8)
        assert emu.e264emu_dbkparam_raw(pkt, got.ctypes.data) == 0
        bs = orc.frame_bs(pkt, n).reshape(n, 32)
        assert np.array_equal(got[:, :32], bs), f"{name} frame {ft}: bS differs at macroblocks {np.nonzero((got[:, :32] != bs).any(1))[0][:8].tolist()}"
        assert np.array_equal(got[:, 32:], expected_ab(pk, w)), f"{name} frame {ft}: alpha / beta / indexA differ"
        # what leaves for memory: the sixteen 8-byte pieces of the deblocking lanes' layout + the macroblock's 16 bytes of beta.  The kernel builds them four slots at a time
        # (byte permutes); the definition is dbkp_piece, slot by slot, on the raw record that has just been checked
        pieces = np.full((n, 144), 0x5A, np.uint8)  # E264_DBK_BYTES
        raw2 = np.zeros((n, 64), np.uint8)
        assert emu.e264emu_dbkparam_frame2(pkt, pieces.ctypes.data, raw2.ctypes.data) == 0
        assert np.array_equal(raw2, got)
        exp = np.zeros((n, 144), np.uint8)
        for a in range(n):
            emu.e264emu_dbk_pieces(got[a].ctypes.data, exp[a].ctypes.data)
        assert np.array_equal(pieces, exp), f"{name} frame {ft}: pieces differ at macroblocks {np.nonzero((pieces != exp).any(1))[0][:8].tolist()}"
        # the kernel's small form (e264_dbkparam2_kernel<false>: no room for list 1 in LDS, eight workgroups per CU): what the launcher picks for batches whose
        # validated packets do not predict from list 1 -- every I and P picture: the same records
        if ft != "B":
            p2 = np.full((n, 144), 0xA5, np.uint8)
            r2 = np.zeros((n, 64), np.uint8)
            assert emu.e264emu_dbkparam_frame2_nol1(pkt, p2.ctypes.data, r2.ctypes.data) == 0
            assert np.array_equal(r2, got) and np.array_equal(p2, exp), f"{name} frame {ft}: the small form differs"

