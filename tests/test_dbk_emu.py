"""e264_deblock_kernel's source run on the HOST (tests/emu) against the CPU oracle: the picture the oracle reconstructs
without deblocking goes through the parameter kernel's and the filter kernel's phases, lane by lane, and must come out as
the oracle's deblocked picture -- every macroblock kind, bS 0..4, both transforms, slices with deblocking off / across
slice edges off, filter offsets, frames narrower than a group of 4 and taller than one wave's five rows."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from edge264_amd import packet as P, synth
from oracle.pyoracle import Oracle, _dpb_array

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    d = os.path.join(HERE, "emu")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(d, name))
    lib.e264emu_dbkparam_frame.argtypes = [C.c_char_p, C.c_void_p]
    lib.e264emu_dbkparam_frame.restype = C.c_int
    lib.e264emu_deblock_frame2.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.e264emu_deblock_frame2.restype = C.c_int
    return lib


@pytest.fixture(scope="module", params=["groups_of_4", "groups_of_2"])
def emu(request):
    """the product build (strips of eight macroblocks, fetch / flush groups of four) and -DE264_DBK_GS=2 (strips of four: the
    twelve-wave variant measured in round 5, profiles/r05_ablations.txt item 1)"""
    lib = _load("libe264_pred_emu.so" if request.param == "groups_of_4" else "libe264_pred_emu_gs2.so")
    lib.has_zeroskip = request.param == "groups_of_2"
    return lib


CASES = [
    ("ipb", "IPB", 9, 7, dict()),
    ("one_mb", "IP", 1, 1, dict()),
    ("narrow", "IPB", 3, 11, dict(num_refs=2)),                 # less than a group of 4 wide, three waves tall
    ("wide", "IPP", 21, 6, dict(residual_prob=0.9)),            # the 8-macroblock strips wrap twice
    ("slices_idc2", "IPBP", 11, 6, dict(slices_per_frame=4, deblock_idc=2, filter_offsets=(6, -4))),
    ("t8x8_pcm", "IPB", 7, 5, dict(t8x8=True, pcm_prob=0.2, i_kinds=(P.MB_I4x4, P.MB_I8x8, P.MB_I16x16))),
    ("intra_in_inter", "IPP", 10, 12, dict(intra_in_inter=0.4, filter_offsets=(-6, 6))),
    ("smooth", "IPP", 8, 6, dict(residual_prob=0.1, p_skip=0.5)),
    ("two_wide", "IPB", 2, 6, dict()),                          # one chroma piece (two macroblocks) is the whole row
    ("five_wide", "IPP", 5, 3, dict(residual_prob=0.8)),        # a group of four + one: the chroma piece of the last macroblock hangs over the row
    ("six_wide", "IPB", 6, 4, dict(num_refs=2)),
    ("tall", "IPP", 5, 35, dict(residual_prob=0.6)),            # 7 mixed groups / 5 luma groups / 3 chroma groups: every hand-off between waves
    ("h17", "IPB", 4, 17, dict(intra_in_inter=0.3)),            # one row into the second chroma group, the third luma group
    # P pictures whose macroblocks have no edge to filter at all (one reference, zero vectors, no residual, no intra): whole steps take
    # the copy-only path (dk_vcopy); with a few coded macroblocks the two paths alternate inside a group
    ("static", "IPP", 13, 10, dict(num_refs=1, mv_range=0, residual_prob=0.0, intra_in_inter=0.0)),
    ("static_some_coded", "IPPP", 12, 20, dict(num_refs=1, mv_range=0, residual_prob=0.04, intra_in_inter=0.01)),
]


@pytest.mark.parametrize("split", [0, 1], ids=["mixed_waves", "luma_and_chroma_waves"])
@pytest.mark.parametrize("name,gop,w,h,kw", CASES, ids=[c[0] for c in CASES])
def test_deblock_emu(emu, name, gop, w, h, kw, split):
    g = synth.StreamSynth(w, h, seed=len(name) * 7 + 1, **kw)
    nb = P.frame_bytes(w, h)
    rng = np.random.default_rng(5)
    dpb = [rng.integers(0, 256, nb + 16, dtype=np.uint8) for _ in range(6)] + [None] * 26
    orc = Oracle()
    for i, ft in enumerate(gop):
        pkt = g.next_frame(ft)
        d = int(P.Packet(pkt).hdr["dst_slot"])
        mine = [None if b is None else b.copy() for b in dpb]
        orc.decode_frame(pkt, mine, 1)          # reconstruction only
        prm = np.zeros(w * h * 146 + 64, np.uint8)  # E264_SCRATCH_BYTES (edge264_amd/csrc/e264_kernels.h): 144 bytes of parameters per macroblock + the intra bitmap
        assert emu.e264emu_dbkparam_frame(pkt, prm.ctypes.data) == 0
        assert emu.e264emu_deblock_frame2(pkt, _dpb_array(mine), prm.ctypes.data, split) == 0
        orc.decode_frame(pkt, dpb, 3)           # reconstruction + deblocking: the reference for this frame and the next
        sY = w * 16
        got_y = mine[d][:sY * h * 16].reshape(h * 16, sY)
        exp_y = dpb[d][:sY * h * 16].reshape(h * 16, sY)
        bad = got_y != exp_y
        assert not bad.any(), f"{name} frame {i} ({ft}): luma differs at (y, x) {np.argwhere(bad)[:6].tolist()}"
        got_c = mine[d][sY * h * 16:nb].reshape(h * 8, sY)
        exp_c = dpb[d][sY * h * 16:nb].reshape(h * 8, sY)
        badc = got_c != exp_c
        assert not badc.any(), f"{name} frame {i} ({ft}): chroma differs at (y, x) {np.argwhere(badc)[:6].tolist()}"
        assert np.array_equal(mine[d][nb:], dpb[d][nb:])
    if name.startswith("static"):  # the variant build has the copy-only path (-DE264_DBK_ZEROSKIP=1): both paths ran there
        z, f = C.c_long(), C.c_long()
        emu.e264emu_deblock_step_counts(C.byref(z), C.byref(f), 1)
        assert f.value > 0 and (z.value > 0) == emu.has_zeroskip, (z.value, f.value)


# ---- the packed edge arithmetic against the standard's formulas (8.7.2.3 / 8.7.2.4), line by line ----------------------
TC0 = [[0] * 52,
       [0] * 23 + [1] * 10 + [2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13],
       [0] * 21 + [1] * 10 + [2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 8, 8, 10, 11, 12, 13, 15, 17],
       [0] * 17 + [1] * 10 + [2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25]]


def clip3(lo, hi, v):
    return min(max(v, lo), hi)


def spec_edge(s, i, bS, alpha, beta, ia, chroma):
    """filter the edge between s[i-1] and s[i] of the sample list s, in place"""
    if bS == 0:
        return
    p = [s[i - 1 - k] for k in range(4)]
    q = [s[i + k] for k in range(4)]
    if not (abs(p[0] - q[0]) < alpha and abs(p[1] - p[0]) < beta and abs(q[1] - q[0]) < beta):
        return
    ap, aq = abs(p[2] - p[0]) < beta, abs(q[2] - q[0]) < beta
    P_, Q_ = list(p), list(q)
    if bS < 4:
        tc0 = TC0[bS][ia]
        tc = tc0 + 1 if chroma else tc0 + int(ap) + int(aq)
        delta = clip3(-tc, tc, (((q[0] - p[0]) << 2) + (p[1] - q[1]) + 4) >> 3)
        P_[0] = clip3(0, 255, p[0] + delta)
        Q_[0] = clip3(0, 255, q[0] - delta)
        if not chroma and ap:
            P_[1] = p[1] + clip3(-tc0, tc0, (p[2] + ((p[0] + q[0] + 1) >> 1) - (p[1] << 1)) >> 1)
        if not chroma and aq:
            Q_[1] = q[1] + clip3(-tc0, tc0, (q[2] + ((p[0] + q[0] + 1) >> 1) - (q[1] << 1)) >> 1)
    else:
        small = abs(p[0] - q[0]) < (alpha >> 2) + 2
        if not chroma and ap and small:
            P_[0] = (p[2] + 2 * p[1] + 2 * p[0] + 2 * q[0] + q[1] + 4) >> 3
            P_[1] = (p[2] + p[1] + p[0] + q[0] + 2) >> 2
            P_[2] = (2 * p[3] + 3 * p[2] + p[1] + p[0] + q[0] + 4) >> 3
        else:
            P_[0] = (2 * p[1] + p[0] + q[1] + 2) >> 2
        if not chroma and aq and small:
            Q_[0] = (p[1] + 2 * p[0] + 2 * q[0] + 2 * q[1] + q[2] + 4) >> 3
            Q_[1] = (p[0] + q[0] + q[1] + q[2] + 2) >> 2
            Q_[2] = (2 * q[3] + 3 * q[2] + q[1] + q[0] + p[0] + 4) >> 3
        else:
            Q_[0] = (2 * q[1] + q[0] + p[1] + 2) >> 2
    for k in range(4):
        s[i - 1 - k] = P_[k]
        s[i + k] = Q_[k]


ALPHA = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255]
BETA = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18]


def test_edge_arithmetic_vs_standard(emu):
    emu.e264emu_dk_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    emu.e264emu_dk_filter.restype = None
    rng = np.random.default_rng(11)
    n_changed = n_strong = 0
    for trial in range(3000):
        prm = np.zeros(64, np.uint8)
        prm[:32] = rng.choice([0, 1, 2, 3, 4], 32, p=[0.15, 0.2, 0.2, 0.2, 0.25])
        bs = prm[:32].reshape(2, 4, 4)
        bs[:, 1:] = np.minimum(bs[:, 1:], 3)  # bS 4 only exists on macroblock edges
        for i in range(9):
            ia = int(rng.integers(16, 52))
            prm[32 + i], prm[41 + i], prm[50 + i] = ALPHA[ia], BETA[min(max(ia + int(rng.integers(-6, 7)), 0), 51)], ia
        lane = int(rng.integers(0, 12)) + 12 * int(rng.integers(0, 5))
        dirn = int(rng.integers(0, 2))
        # smooth lines with small steps, so that every branch of the filter is taken
        base = rng.integers(0, 256)
        lines = np.clip(base + np.cumsum(rng.integers(-3, 4, (2, 20)), axis=1) + rng.integers(-2, 3, (2, 20)) * (trial % 3 == 0), 0, 255).astype(np.uint8)
        got = lines.copy()
        emu.e264emu_dk_filter(got.ctypes.data, prm.ctypes.data, lane, dirn)
        r = lane % 12
        chroma = r >= 8
        pi = r - 8 if chroma else r
        seg = pi if chroma else pi >> 1
        exp = lines.astype(int)
        for ln in range(2):
            s = exp[ln].tolist()
            for e in range(4):
                if chroma:
                    bS = int(prm[dirn * 16 + (e & 1) * 8 + seg])
                    abi = (1 + (e >> 1)) * 3 + (0 if e & 1 else 1 + dirn)
                    # only p1 p0 | q0 q1 of the slot are samples of this plane: the standard's chroma filter reads nothing else
                    win = s[4 * e + 2:4 * e + 6]
                    t = [0, 0] + win + [0, 0]
                    spec_edge(t, 4, bS, int(prm[32 + abi]), int(prm[41 + abi]), int(prm[50 + abi]), True)
                    s[4 * e + 2:4 * e + 6] = t[2:6]
                else:
                    bS = int(prm[dirn * 16 + e * 4 + seg])
                    abi = 1 + dirn if e == 0 else 0
                    spec_edge(s, 4 + 4 * e, bS, int(prm[32 + abi]), int(prm[41 + abi]), int(prm[50 + abi]), False)
                    n_strong += bS == 4
            exp[ln] = s
        if chroma:  # taps outside the four windows are not samples: whatever they hold must come back unchanged
            pass
        assert np.array_equal(got.astype(int), exp), (trial, lane, dirn, prm[:32].tolist(), lines.tolist(), got.tolist(), exp.tolist())
        n_changed += int((got != lines).sum())
    assert n_changed > 20000 and n_strong > 300
