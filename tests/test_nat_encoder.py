"""The encoder behind the encoder-shaped fixtures (tests/golden/nat_encoder.py) -- TEST INFRASTRUCTURE, checked like the rest of it: its sample
interpolation (the 16 quarter-sample planes it searches on, its bilinear chroma) against the CPU oracle's restatement of
/root/reference/src/edge264_inter.c:416-1091, its transform pair against itself, its vector prediction on a hand-made neighbourhood, and what the
committed streams look like (the reason they exist: skip runs and coherent motion, reference_md5.json `encoder_stats`).  The streams themselves are
pinned like every other fixture: md5 of every picture from the unmodified reference decoder (tests/test_frontend_capture.py, test_frontend_hip.py)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

try:
    import nat_encoder as ne  # imports make_streams, which only needs /root/reference when it RUNS the writers
except Exception as e:  # noqa: BLE001
    pytest.skip(f"nat_encoder not importable: {e}", allow_module_level=True)

from oracle.pyoracle import Oracle, oracle_chroma_mc, oracle_luma_mc  # noqa: E402


def test_quarter_sample_planes_match_the_oracle():
    rng = np.random.default_rng(3)
    H, W = 48, 64
    Y = rng.integers(0, 256, (H, W), dtype=np.uint8)
    Y[:8] = rng.integers(0, 2, (8, W)) * 255  # extremes: the six-tap sums leave 0..255 and are clipped
    planes = ne.qpel_planes(ne.pad(Y))
    o = Oracle()
    flat = np.ascontiguousarray(Y)
    for fy in range(4):
        for fx in range(4):
            for (x, y, bw, bh) in ((16, 16, 16, 16), (8, 24, 8, 8), (0, 0, 16, 16), (W - 16, H - 16, 16, 16)):
                for (ix, iy) in ((0, 0), (3, -2), (-5, 4)):
                    mv = (4 * ix + fx, 4 * iy + fy)
                    want = oracle_luma_mc(o, flat, W, W, H, x, y, mv[0], mv[1], bw, bh)
                    got = ne.luma_pred(planes, x, y, mv, bw, bh)
                    assert np.array_equal(got, want), (fx, fy, x, y, ix, iy)


def test_chroma_prediction_matches_the_oracle():
    rng = np.random.default_rng(4)
    H, W = 24, 32
    C = rng.integers(0, 256, (H, W), dtype=np.uint8)
    o = Oracle()
    P = ne.pad(C)
    for mvx in (-19, -8, -3, 0, 5, 7, 12, 33):
        for mvy in (-10, -1, 0, 4, 6, 17):
            for (x, y, n) in ((8, 8, 8), (0, 0, 4), (W - 8, H - 8, 8), (12, 4, 4)):
                want = oracle_chroma_mc(o, np.ascontiguousarray(C), W, W, H, x, y, mvx, mvy, n, n)
                got = ne.chroma_pred(P, x, y, (mvx, mvy), n, n)
                assert np.array_equal(got, want), (mvx, mvy, x, y, n)


def test_transform_pair_reconstructs_within_the_quantiser_step():
    rng = np.random.default_rng(5)
    for qp in (20, 28, 30, 36):
        res = rng.integers(-60, 61, (50, 4, 4))
        lev = ne.quant(ne.fwd4x4(res), qp, False)
        rec = ne.inv4x4(ne.dequant(lev, qp))
        step = 0.625 * 2 ** (qp / 6)
        assert np.abs(rec - res).max() <= 2.5 * step + 1, qp   # sixteen coefficients' rounding errors (dead zone: up to a step each) land on one sample
        assert np.abs(rec - res).mean() < step / 2, qp
    assert not ne.quant(ne.fwd4x4(np.zeros((4, 4), np.int64)), 28, False).any()


def test_8x8_transform_pair_reconstructs_within_the_quantiser_step():
    """the High-profile variant's forward 8x8 transform + quantiser against the standard's dequantiser + inverse (as the decoder runs them)"""
    rng = np.random.default_rng(6)
    for qp in (20, 28, 31, 36, 40):
        res = rng.integers(-60, 61, (40, 8, 8))
        rec = ne.inv8x8(ne.dequant8(ne.quant8(ne.fwd8x8(res), qp, False), qp))
        step = 0.625 * 2 ** (qp / 6)
        assert np.abs(rec - res).mean() < step / 2, qp
    ramp = np.add.outer(np.arange(8), np.arange(8)) * 3          # a smooth block: two or three low-frequency levels describe it
    lev = ne.quant8(ne.fwd8x8(ramp), 28, False)
    assert (lev != 0).sum() <= 4 and np.abs(ne.inv8x8(ne.dequant8(lev, 28)) - ramp).max() <= 6
    # the zig-zag of an 8x8 block is a permutation and starts along the first anti-diagonals
    assert sorted(ne.ZZ8) == list(range(64)) and ne.ZZ8[:6] == [0, 1, 8, 16, 9, 2]


def test_vector_prediction_rules():
    m = ne.Motion(4, 3)
    # nothing decoded yet: everything unavailable -> (0, 0); P_Skip without a left / top neighbour -> (0, 0)
    assert m.mvp(0, 4, 4, 4, 0) == (0, 0) and m.p_skip_mv(0, 0) == (0, 0)
    m.set(0, 0, 0, 4, 4, 0, (8, -4))      # macroblock (0, 0)
    m.set(0, 4, 0, 4, 4, 0, (12, 0))      # (1, 0)
    m.set(0, 8, 0, 4, 4, -1)              # (2, 0) intra
    m.set(0, 0, 4, 4, 4, 0, (4, 4))       # (0, 1)
    # macroblock (1, 1): A = (0, 1) (4, 4), B = (1, 0) (12, 0), C = (2, 0) intra: available, refIdx -1 -> median of (4,4) (12,0) (0,0)
    assert m.mvp(0, 4, 4, 4, 0) == (4, 0)
    # only one neighbour with the wanted reference: its vector
    m.set(0, 4, 0, 4, 4, -1)
    assert m.mvp(0, 4, 4, 4, 0) == (4, 4)
    # P_Skip: a neighbour with reference 0 and a zero vector forces (0, 0)
    m.set(0, 0, 4, 4, 4, 0, (0, 0))
    assert m.p_skip_mv(4, 4) == (0, 0)
    # top row: B and C unavailable, A available -> A's vector
    m2 = ne.Motion(4, 3)
    m2.set(0, 0, 0, 4, 4, 0, (-6, 2))
    assert m2.mvp(0, 4, 0, 4, 0) == (-6, 2)


def test_the_committed_streams_are_encoder_shaped():
    with open(os.path.join(HERE, "golden", "streams", "reference_md5.json")) as f:
        sums = json.load(f)
    for name, min_skip in (("nat1080_ipp30", 0.40), ("cabac_nat1080_ibbp30", 0.45)):
        st = sums[name]["encoder_stats"]
        mbs = 120 * 68 * len(sums[name]["frames"])
        assert st["skip"] / mbs > min_skip, (name, st)          # long skip runs
        assert st["intra"] - 120 * 68 < 0.03 * mbs, (name, st)  # intra almost only in the I picture
        assert st["p8x8"] < 0.1 * mbs and st["coded"] < 0.4 * mbs, (name, st)
        size = os.path.getsize(os.path.join(HERE, "golden", "streams", name + ".264"))
        assert 4e6 < size * 8 * 30 / len(sums[name]["frames"]) < 6e6, (name, size)  # 4 - 6 Mbit/s at 30 pictures/s
    assert sums["cabac_nat1080_ibbp30"]["encoder_stats"]["direct"] > 20000   # B_Direct_16x16 beside B_Skip


def test_the_rate_control_shaped_streams_vary_qp_and_cut_slices():
    """nat_small_aq_slices_* (adaptive quantisation + a slice every few macroblock rows): what reaches the kernels really has several QPs inside a
    picture, edges between macroblocks of different QP, several slice entries per picture and -- in the CABAC one -- disable_deblocking_filter_idc 2."""
    from edge264_amd import front, packet as P
    for name, rows, idc in (("nat_small_aq_slices_ipp8", 4, 0), ("cabac_nat_small_aq_slices_ibbp10", 5, 2)):
        data = open(os.path.join(HERE, "golden", "streams", name + ".264"), "rb").read()
        packets, _, _ = front.capture_packets(data)
        assert len(packets) == (8 if idc == 0 else 10)
        for pkt in packets:
            pk = P.Packet(pkt)
            W, H = pk.width_mbs, pk.height_mbs
            qp = pk.mbs["qp"][:, 0].reshape(H, W).astype(int)
            assert qp.max() - qp.min() >= 4, (name, qp.min(), qp.max())
            assert (np.diff(qp, axis=1) != 0).mean() > 0.15, name             # neighbouring macroblocks of different QP: qPav, not QP, on the edge
            sl = pk.mbs["slice"].reshape(H, W).astype(int)
            assert (sl[:, 0] == sl[:, -1]).all() and len(np.unique(sl)) == -(-H // rows), (name, np.unique(sl))
            assert all(int(s["disable_deblocking_filter_idc"]) == idc for s in pk.slices[:len(np.unique(sl))])


def test_the_fade_streams_carry_explicit_weights_that_follow_the_fade():
    """nat_small_fade_wp_* (a fade towards black): every P slice reaches the kernels with explicit luma weights below 1.0 (denominator 2^5) for
    reference 0 -- the picture is darker than what it predicts from -- and most macroblocks still skip or predict without residual."""
    from edge264_amd import front, packet as P
    with open(os.path.join(HERE, "golden", "streams", "reference_md5.json")) as f:
        sums = json.load(f)
    for name in ("nat_small_fade_wp_ipp8", "cabac_nat_small_fade_wp_ipp8"):
        data = open(os.path.join(HERE, "golden", "streams", name + ".264"), "rb").read()
        packets, _, _ = front.capture_packets(data)
        assert len(packets) == 8
        for pkt in packets[1:]:
            s = P.Packet(pkt).slices[0]
            assert int(s["slice_type"]) == 0 and int(s["luma_log2_weight_denom"]) == 5
            w = int(s["explicit_weights"][0][0])
            assert 16 <= w < 32, (name, w)
        assert sums[name]["encoder_stats"]["skip"] > 0.2 * 240 * 7, sums[name]["encoder_stats"]


def test_directional_vector_prediction_of_rectangular_partitions():
    """8.4.1.3: the upper 16x8 partition predicts from B, the lower from A, the left 8x16 from A, the right from C -- when that neighbour uses the
    same reference; the median otherwise."""
    m = ne.Motion(3, 2)
    m.set(0, 0, 0, 4, 4, 0, (8, 0))       # macroblock (0, 0)
    m.set(0, 4, 0, 4, 4, 1, (0, 12))      # (1, 0): another reference
    m.set(0, 8, 0, 4, 4, 0, (-4, -4))     # (2, 0)
    m.set(0, 0, 4, 4, 4, 0, (2, 6))       # (0, 1)
    # macroblock (1, 1): A = (2, 6) ref 0, B = (0, 12) ref 1, C = (-4, -4) ref 0
    assert m.mvp(0, 4, 4, 4, 1, "B") == (0, 12)           # upper 16x8 with reference 1: B's vector
    assert m.mvp(0, 4, 4, 4, 0, "B") == m.mvp(0, 4, 4, 4, 0) == (0, 6)  # B has another reference: median of (2,6) (0,12) (-4,-4)
    assert m.mvp(0, 4, 6, 4, 0, "A") == (2, 6)            # lower 16x8: A's vector
    assert m.mvp(0, 4, 4, 2, 0, "A") == (2, 6)            # left 8x16
    assert m.mvp(0, 6, 4, 2, 0, "C") == (-4, -4)          # right 8x16: C = macroblock (2, 0)
    # lower 16x8 whose A has another reference: the median, with C replaced by D (the block up-left, C lies in a macroblock not decoded yet)
    m.set(0, 4, 4, 4, 2, 1, (10, 10))                     # the upper partition is set first
    assert m.mvp(0, 4, 6, 4, 1, "A") == (10, 10)          # A (2,6) ref 0, B = upper partition ref 1, D = (0,1) ref 0: only B has reference 1


def test_the_rectangular_partition_streams_use_them():
    with open(os.path.join(HERE, "golden", "streams", "reference_md5.json")) as f:
        sums = json.load(f)
    for name in ("nat_small_rect_ipp8", "cabac_nat_small_rect_ipp8"):
        st = sums[name]["encoder_stats"]
        assert st["p16x8"] > 150 and st["p8x16"] > 150 and st["p16"] > st["p16x8"] + st["p8x16"], (name, st)  # a minority, as encoders choose them


def test_the_sub_partition_streams_use_them():
    with open(os.path.join(HERE, "golden", "streams", "reference_md5.json")) as f:
        sums = json.load(f)
    for name in ("nat_small_sub_ipp8", "cabac_nat_small_sub_ipp8"):
        st = sums[name]["encoder_stats"]
        assert 250 < st["subparts"] <= st["p8x8"], (name, st)   # P_8x8 macroblocks with a quadrant split below 8x8


def test_the_intra8x8_streams_use_it():
    """nat_small_i8x8_*: Intra8x8 macroblocks chosen by the encoder (transform_size_8x8_flag on an I_NxN macroblock) reach the kernels as such."""
    from edge264_amd import front, packet as P
    with open(os.path.join(HERE, "golden", "streams", "reference_md5.json")) as f:
        sums = json.load(f)
    for name in ("nat_small_i8x8_iipp6", "cabac_nat_small_i8x8_iipp6"):
        assert sums[name]["encoder_stats"]["i8x8"] > 80, sums[name]["encoder_stats"]
        data = open(os.path.join(HERE, "golden", "streams", name + ".264"), "rb").read()
        packets, _, _ = front.capture_packets(data)
        n8 = sum(int((P.Packet(p).mbs["kind"] == P.MB_I8x8).sum()) for p in packets)
        assert n8 == sums[name]["encoder_stats"]["i8x8"], (name, n8)
