"""e264_pred_kernel's source run on the HOST (tests/emu) against the CPU oracle: every inter / PCM macroblock of synthetic
P and B frames (all partition shapes, all 16 quarter-sample positions, vectors far outside the frame, both transforms,
scaling lists, every weighting scheme, PCM, multiple slices).  Finds logic errors without a GPU; the -m gpu tests run the
same comparison through the C-ABI on the device."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from edge264_amd import packet as P, synth
from oracle.pyoracle import Oracle, _dpb_array

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", params=["product", "paired_tiles"])
def emu(request):
    """product: the default build; paired_tiles: the variant build of tests/emu (-DE264_PRED_PAIR_TILES=1: the chroma of a pair of quadrants fetched by
    plane on tiles of one-partition macroblocks -- measured and not the default, kept in the tree, so it stays tested)"""
    d = os.path.join(HERE, "emu")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(d, "libe264_pred_emu.so" if request.param == "product" else "libe264_pred_emu_gs2.so"))
    lib.e264emu_pred_frame.argtypes = [C.c_char_p, C.c_void_p]
    lib.e264emu_pred_frame.restype = C.c_int
    return lib


def mb_mask(pkt, w, h):
    """boolean planes (luma, chroma) of the samples e264_pred_kernel owns: inter and PCM macroblocks"""
    pk = P.Packet(pkt)
    kinds = pk.mbs["kind"].reshape(h, w)
    own = (kinds == P.MB_INTER) | (kinds == P.MB_PCM)
    return np.kron(own, np.ones((16, 16), bool)), np.kron(own, np.ones((8, 8), bool))


CASES = {
    "p_basic": dict(gop="IPP", w=5, h=4, kw=dict()),
    "p_wide_tiles": dict(gop="IPP", w=37, h=19, kw=dict(num_refs=2)),
    "b_default": dict(gop="IPBB", w=18, h=9, kw=dict()),
    "b_implicit": dict(gop="IPBB", w=7, h=5, kw=dict(weighted=2)),
    "pb_explicit": dict(gop="IPBPB", w=7, h=5, kw=dict(weighted=1)),
    "t8x8_scaling": dict(gop="IPB", w=9, h=6, kw=dict(t8x8=True, scaling=True, residual_prob=0.9)),
    "stress_far_mvs": dict(gop="IPBP", w=4, h=3, kw=dict(stress=True, mv_range=400, residual_prob=0.8)),
    "pcm_slices": dict(gop="IPB", w=6, h=5, kw=dict(pcm_prob=0.3, intra_in_inter=0.3, slices_per_frame=3)),
    # tiles made of one-partition macroblocks (P_Skip / 16x16, one vector per list): the chroma of a pair of quadrants fetched by PLANE (pred_pair_chroma:
    # at least three quarters of a tile's inter macroblocks), with vectors far outside the frame, weights and both lists
    "uni_tiles_p": dict(gop="IPP", w=20, h=6, kw=dict(p_skip=0.95, num_refs=2, residual_prob=0.2)),
    "uni_tiles_far": dict(gop="IPBP", w=5, h=4, kw=dict(p_skip=0.95, stress=True, mv_range=400)),
    "uni_tiles_b_weighted": dict(gop="IPBB", w=18, h=5, kw=dict(p_skip=0.95, weighted=1)),
    "uni_tiles_b_implicit": dict(gop="IPBB", w=7, h=5, kw=dict(p_skip=0.95, weighted=2)),
    "all_residual": dict(gop="IPP", w=6, h=4, kw=dict(residual_prob=1.0, p_skip=0.0)),
    # QP walks around 36: lanes of one wave on both sides of the 8x8 dequantisation's two forms (residual.c:214-247; one flow since round 4)
    "t8x8_qp_around_36": dict(gop="IPB", w=9, h=6, kw=dict(t8x8=True, scaling=True, residual_prob=1.0, qp_base=37)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_pred_emu_vs_oracle(emu, name):
    c = CASES[name]
    w, h = c["w"], c["h"]
    for seed in (1, 2):
        g = synth.StreamSynth(w, h, seed=seed * 77 + len(name), **c["kw"])
        nb = P.frame_bytes(w, h)
        rng = np.random.default_rng(seed)
        dpb = [rng.integers(0, 256, nb + 16, dtype=np.uint8) for _ in range(6)] + [None] * 26
        orc = Oracle()
        for ft in c["gop"]:
            pkt = g.next_frame(ft)
            d = int(P.Packet(pkt).hdr["dst_slot"])
            mine = [None if b is None else b.copy() for b in dpb]
            before = mine[d].copy()
            orc.decode_frame(pkt, dpb, 1)  # reconstruction only: what the prediction kernel + intra kernel produce
            assert emu.e264emu_pred_frame(pkt, _dpb_array(mine)) == 0
            my, mc = mb_mask(pkt, w, h)
            sY, sC = w * 16, w * 16
            got_y = mine[d][:sY * h * 16].reshape(h * 16, sY)
            exp_y = dpb[d][:sY * h * 16].reshape(h * 16, sY)
            bad = (got_y != exp_y) & my
            assert not bad.any(), f"{name} seed {seed} frame {ft}: luma differs at (y, x) {np.argwhere(bad)[:5].tolist()}"
            got_c = mine[d][sY * h * 16:nb].reshape(h * 8, sC)
            exp_c = dpb[d][sY * h * 16:nb].reshape(h * 8, sC)
            mcc = np.concatenate([mc, mc], axis=1)  # a chroma row is [Cb | Cr]
            badc = (got_c != exp_c) & mcc
            assert not badc.any(), f"{name} seed {seed} frame {ft}: chroma differs at (y, x) {np.argwhere(badc)[:5].tolist()}"
            # samples the kernel does not own (intra, absent macroblocks) stay untouched
            assert np.array_equal(got_y[~my], before[:sY * h * 16].reshape(h * 16, sY)[~my])
            assert np.array_equal(got_c[~mcc], before[sY * h * 16:nb].reshape(h * 8, sC)[~mcc])
