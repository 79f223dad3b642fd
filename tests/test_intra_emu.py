"""e264_intra_kernel's source (edge264_amd/csrc/e264_intra.h) run on the HOST as it is (tests/emu/intra_emu.cpp: the 64 lanes of a
wave as fibres that meet at the collectives), after the prediction kernel's, against the CPU oracle: whole pictures -- Intra4x4 / 8x8 / 16x16 with every resolved mode the generator can draw,
both transforms, scaling lists, int8 and int16 levels, DC-only blocks, uncoded macroblocks, intra macroblocks inside P / B
pictures, several slices.  Finds logic errors without a GPU; the -m gpu tests run the same comparison through the C-ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from edge264_amd import packet as P, synth
from oracle.pyoracle import Oracle, _dpb_array

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(HERE, "emu")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(d, "libe264_pred_emu.so"))
    ilib = C.CDLL(os.path.join(d, "libe264_intra_emu.so"))
    lib.e264emu_intra_frame = ilib.e264emu_intra_frame
    lib.e264emu_intra_frame2 = ilib.e264emu_intra_frame2
    lib.e264emu_intra_frame_planes = ilib.e264emu_intra_frame_planes
    for fn in (lib.e264emu_pred_frame, lib.e264emu_intra_frame, lib.e264emu_intra_frame_planes):
        fn.argtypes = [C.c_char_p, C.c_void_p]
        fn.restype = C.c_int
    for fn in (lib.e264emu_pred_frame2, lib.e264emu_intra_frame2):
        fn.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
        fn.restype = C.c_int
    return lib


def scratch_bytes(n_mbs):
    return n_mbs * 146 + 64  # edge264_amd/csrc/e264_kernels.h E264_SCRATCH_BYTES


CASES = {
    "i_all_kinds": dict(gop="II", w=9, h=6, kw=dict(i_kinds=(P.MB_I4x4, P.MB_I8x8, P.MB_I16x16), t8x8=True)),
    "i_4x4_only": dict(gop="I", w=11, h=5, kw=dict(i_kinds=(P.MB_I4x4,))),
    "i_16x16_only": dict(gop="I", w=7, h=5, kw=dict(i_kinds=(P.MB_I16x16,))),
    "i_8x8_scaling": dict(gop="II", w=8, h=5, kw=dict(i_kinds=(P.MB_I8x8, P.MB_I4x4), t8x8=True, scaling=True)),
    "i_wide": dict(gop="I", w=70, h=3, kw=dict()),  # more than one 64-macroblock chunk per row
    "i_big_levels": dict(gop="I", w=6, h=4, kw=dict(big_levels=True)),
    "p_with_intra": dict(gop="IPB", w=9, h=7, kw=dict(intra_in_inter=0.4, pcm_prob=0.1, slices_per_frame=3)),
    "i_slices_qp": dict(gop="II", w=8, h=6, kw=dict(slices_per_frame=4, scaling=True)),
    "p_sparse_intra_wide": dict(gop="IPP", w=70, h=5, kw=dict(intra_in_inter=0.01)),  # rows and 64-macroblock chunks without any intra macroblock
    "i_8x8_qp_around_36": dict(gop="II", w=8, h=5, kw=dict(i_kinds=(P.MB_I8x8,), t8x8=True, scaling=True, qp_base=37)),  # both forms of the 8x8 dequantisation
}


def _synth(w, h, seed, kw):
    kw = dict(kw)
    big = kw.pop("big_levels", False)
    g = synth.StreamSynth(w, h, seed=seed, **kw)
    if big:  # levels beyond a signed byte: the int16 layout of the payload
        orig = g._levels

        def levels(n, qp, maxnz, lowfreq):
            c = orig(n, qp, maxnz, lowfreq)
            nzp = np.flatnonzero(c)
            if len(nzp):
                c[nzp[0]] = 300 if c[nzp[0]] > 0 else -300
            return c
        g._levels = levels
    return g


@pytest.mark.parametrize("bitmap", [False, True, "planes"], ids=["scan", "bitmap", "planes"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_intra_emu_vs_oracle(emu, name, bitmap):
    """bitmap: the prediction kernel leaves the intra bitmap in the stream's scratch and the intra kernel skips by it (what a submission does on
    the device); scan: no scratch, every chunk of every row is scanned (the fallback); planes: e264_intra_planes_kernel -- the picture's chroma by one
    workgroup, then its luma by another (the launcher's form for I pictures split off a mixed submission)"""
    planes = bitmap == "planes"
    bitmap = bitmap is True
    c = CASES[name]
    w, h = c["w"], c["h"]
    scratch = np.full(scratch_bytes(w * h), 0xFF if bitmap else 0, np.uint8)  # (stale ones: the kernel must overwrite every entry it reads)
    for seed in (1, 2, 3):
        g = _synth(w, h, seed * 131 + len(name), c["kw"])
        nb = P.frame_bytes(w, h)
        rng = np.random.default_rng(seed)
        dpb = [rng.integers(0, 256, nb + 16, dtype=np.uint8) for _ in range(6)] + [None] * 26
        orc = Oracle()
        for ft in c["gop"]:
            pkt = g.next_frame(ft)
            d = int(P.Packet(pkt).hdr["dst_slot"])
            mine = [None if b is None else b.copy() for b in dpb]
            orc.decode_frame(pkt, dpb, 1)  # reconstruction only: prediction kernel + intra kernel
            if bitmap:
                scratch[w * h * 144:] = 0x00 if ft == "I" else 0xFF  # stale entries of the picture before, both ways
                assert emu.e264emu_pred_frame2(pkt, _dpb_array(mine), scratch.ctypes.data) == 0
                assert emu.e264emu_intra_frame2(pkt, _dpb_array(mine), scratch.ctypes.data) == 0
                ntx = (w + 15) // 16
                bm = scratch[w * h * 144:w * h * 144 + 2 * ntx * h].view("<u2").reshape(h, ntx)
                kinds = P.Packet(pkt).mbs["kind"].reshape(h, w)
                want = np.isin(kinds, (P.MB_I4x4, P.MB_I8x8, P.MB_I16x16))
                got_bits = np.array([[bm[y, x >> 4] >> (x & 15) & 1 for x in range(w)] for y in range(h)], bool)
                assert np.array_equal(got_bits, want), f"{name} frame {ft}: intra bitmap differs"
            else:
                assert emu.e264emu_pred_frame(pkt, _dpb_array(mine)) == 0
                assert (emu.e264emu_intra_frame_planes if planes else emu.e264emu_intra_frame)(pkt, _dpb_array(mine)) == 0
            sY = w * 16
            got_y = mine[d][:sY * h * 16].reshape(h * 16, sY)
            exp_y = dpb[d][:sY * h * 16].reshape(h * 16, sY)
            bad = got_y != exp_y
            assert not bad.any(), f"{name} seed {seed} frame {ft}: luma differs at (y, x) {np.argwhere(bad)[:5].tolist()}"
            got_c = mine[d][sY * h * 16:nb].reshape(h * 8, sY)
            exp_c = dpb[d][sY * h * 16:nb].reshape(h * 8, sY)
            badc = got_c != exp_c
            assert not badc.any(), f"{name} seed {seed} frame {ft}: chroma differs at (y, x) {np.argwhere(badc)[:5].tolist()}"
            dpb[d][:] = mine[d]  # (identical) keep going from the emulated picture


def test_check_vectors_through_the_emulated_kernels(emu):
    """The reference's published vectors (src/edge264_check.c:185-357, tests/golden_packets.py: all 14 + 32 + 7 + 7 intra modes, the 48
    luma quarter-sample positions, chroma) through the kernels' SOURCE on the host: what tests/test_hip_golden.py does on the GPU,
    here without one -- a change to e264_intra.h / e264_pred.h that breaks a mode shows in the CPU suite, no oracle in between."""
    from tests import golden_packets as GP
    nb = P.frame_bytes(GP.W, GP.H)
    n = 0
    for name, pkt, init, checks in GP.cases():
        dpb = [None] * 32
        for s in (GP.DST, GP.REF):
            dpb[s] = np.zeros(nb + 16, np.uint8)
        for s, buf in init.items():
            dpb[s][:nb] = buf[:nb]
        assert emu.e264emu_pred_frame(pkt, _dpb_array(dpb)) == 0, name
        assert emu.e264emu_intra_frame(pkt, _dpb_array(dpb)) == 0, name
        GP.check(name, dpb[GP.DST][:nb], checks)
        n += 1
    assert n == 110
