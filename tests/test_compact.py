"""The WIRE form of a command packet (version 5, include/edge264_compact.h): P_Skip / plain 16x16 macroblocks without residual in 12 bytes instead of 40.
Host side, no GPU: the fold (e264_compact_packet) and its inverse (e264_expand_packet) on every committed stream and on synthetic pictures, the
validation of damaged wire packets, the front end's own wire output, and e264_expand_kernel's source compiled for the host (tests/emu) against the
host expansion byte for byte -- the four kernels then read a wire packet through two pointers (open_frame), which the emulated kernels show too.
The -m gpu part (tests/test_hip_wire.py) sends the same packets through the C-ABI."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

from edge264_amd import backend, front, packet as P, synth
from oracle.pyoracle import Oracle, _dpb_array

HERE = os.path.dirname(os.path.abspath(__file__))
STREAMS = os.path.join(HERE, "golden", "streams")
SMALL = sorted(p for p in glob.glob(os.path.join(STREAMS, "*.264")) if os.path.getsize(p) < 60_000)


def _have_front():
    try:
        front.load()
        return True
    except front.FrontError:
        return False


needs_front = pytest.mark.skipif(not _have_front(), reason="front-end library not built (needs the reference tree)")


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(HERE, "emu")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.DEVNULL)
    pe = C.CDLL(os.path.join(d, "libe264_pred_emu.so"))
    ie = C.CDLL(os.path.join(d, "libe264_intra_emu.so"))
    pe.e264emu_expand.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    pe.e264emu_set_expand.argtypes = [C.c_void_p]
    ie.e264emu_set_expand.argtypes = [C.c_void_p]
    pe.e264emu_pred_frame2.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
    pe.e264emu_dbkparam_frame2.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
    ie.e264emu_intra_frame2.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
    yield pe, ie
    pe.e264emu_set_expand(None)
    ie.e264emu_set_expand(None)


def synth_packets():
    out = []
    for seed, w, h, gop, kw in ((3, 20, 6, "IPP", dict(p_skip=0.95, num_refs=2, residual_prob=0.2)), (4, 37, 5, "IPBB", dict(p_skip=0.9)),
                                (5, 33, 3, "IPP", dict(p_skip=0.5, slices_per_frame=3)), (6, 6, 5, "IPB", dict(pcm_prob=0.3, intra_in_inter=0.3, p_skip=0.6)),
                                (7, 65, 2, "IPP", dict(p_skip=1.0, residual_prob=0.0)), (8, 1, 1, "IP", dict(p_skip=1.0, residual_prob=0.0))):
        g = synth.StreamSynth(w, h, seed=seed, **kw)
        out += [(f"synth{seed}", bytes(g.next_frame(ft))) for ft in gop]
    return out


def test_fold_and_unfold_synthetic():
    folded = 0
    for name, pkt in synth_packets():
        assert backend.packet_check(pkt) == 0
        wire = backend.packet_compact(pkt)
        assert backend.packet_check(wire) == 0, (name, backend.last_error())
        back = backend.packet_expand(wire)
        assert backend.packet_check(back) == 0, (name, backend.last_error())
        assert backend.packet_compact(back) == wire  # the expansion is canonical: folding it again gives the same bytes
        a, b = P.Packet(pkt), P.Packet(back)
        for f in ("kind", "flags", "qp", "nz_mask", "slice", "coded", "dbk_slice"):
            assert np.array_equal(a.mbs[f], b.mbs[f]), (name, f)
        ma, mb = a.motion, b.motion
        assert (ma is None) == (mb is None)
        if ma is not None:
            assert ma.tobytes() == mb.tobytes(), name  # every macroblock's motion, in the expanded form
        assert bytes(a.data[a.payload_off:a.payload_off + int(a.hdr["payload_bytes"])]) == bytes(b.data[b.payload_off:b.payload_off + int(b.hdr["payload_bytes"])])
        n5 = int(np.frombuffer(wire, "<u4", 1, int(np.frombuffer(wire, P.FRAME_HDR, 1)[0]["mbs_off"]))[0])
        folded += n5
        if n5 * 28 > 2000:
            assert len(wire) < len(pkt)
    assert folded > 500


def test_the_expansion_decodes_to_the_same_pictures():
    """the oracle (a version-4 reader) on the original packet and on fold + unfold: same pictures; P.Packet reads a wire packet as its expansion"""
    g = synth.StreamSynth(9, 5, seed=11, p_skip=0.8, num_refs=2)
    nb = P.frame_bytes(9, 5)
    rng = np.random.default_rng(1)
    dpb = [rng.integers(0, 256, nb + 16, dtype=np.uint8) for _ in range(6)] + [None] * 26
    dpb5 = [None if b is None else b.copy() for b in dpb]
    o4, o5 = Oracle(), Oracle()
    for ft in "IPPBP":
        pkt = bytes(g.next_frame(ft))
        o4.decode_frame(pkt, dpb, 3)
        wire = backend.packet_compact(pkt)
        o5.decode_frame(backend.packet_expand(wire) if wire[4] == 5 else wire, dpb5, 3)
        assert P.Packet(wire).mbs.tobytes() == P.Packet(backend.packet_expand(wire)).mbs.tobytes() if wire[4] == 5 else True
        d = int(P.Packet(pkt).hdr["dst_slot"])
        assert np.array_equal(dpb[d][:nb], dpb5[d][:nb])


@needs_front
def test_every_committed_stream_folds_and_unfolds():
    saved = total = 0
    for f in sorted(glob.glob(os.path.join(STREAMS, "*.264"))):
        if os.path.getsize(f) > 400_000:
            continue
        for pkt in front.capture_packets(open(f, "rb").read())[0]:
            pkt = bytes(pkt)
            wire = backend.packet_compact(pkt)
            assert backend.packet_check(wire) == 0, (f, backend.last_error())
            back = backend.packet_expand(wire)
            assert backend.packet_check(back) == 0, (f, backend.last_error())
            assert backend.packet_compact(back) == wire
            total += len(pkt)
            saved += len(pkt) - len(wire)
    assert saved > 0


@needs_front
def test_front_end_emits_the_same_wire_packets():
    """e264front_set_compact(1): what leaves the front end is the fold of what it would have sent (pictures without inter macroblocks leave as version 4)."""
    n5 = 0
    for f in SMALL:
        data = open(f, "rb").read()
        plain = [bytes(p) for p in front.capture_packets(data)[0]]
        wire = [bytes(p) for p in front.capture_packets(data, compact=True)[0]]
        assert len(plain) == len(wire), f
        for a, b in zip(plain, wire):
            assert backend.packet_check(b) == 0
            if b[4] == 5:
                assert b == backend.packet_compact(a), f
                n5 += 1
            else:
                assert a == b and int(P.Packet(a).hdr["n_inter_mbs"]) == 0, f
    front.capture_packets(b"", compact=False)  # (the switch is global to the library: leave it off)
    assert n5 > 50


def _wire_cases():
    out = [(n, backend.packet_compact(p)) for n, p in synth_packets()]
    return [(n, w) for n, w in out if w[4] == 5]


def test_expand_kernel_source_equals_the_host_expansion(emu):
    """e264_expand.h run thread by thread on the host: the record array and the motion section it writes are the bytes of e264_expand_packet"""
    pe, _ = emu
    cases = _wire_cases()
    if _have_front():
        for f in SMALL[::4]:
            cases += [(os.path.basename(f), backend.packet_compact(bytes(p))) for p in front.capture_packets(open(f, "rb").read())[0]]
    n = 0
    for name, wire in cases:
        if wire[4] != 5:
            continue
        back = backend.packet_expand(wire)
        h = np.frombuffer(back, P.FRAME_HDR, 1)[0]
        want = back[int(h["mbs_off"]):int(h["payload_off"])]
        for nt in (256, 64 * 7, 1):
            area = (C.c_uint8 * (len(want) + 64))()
            C.memset(area, 0xA5, len(want) + 64)
            pe.e264emu_expand(wire, area, nt)
            assert bytes(area[:len(want)]) == want, (name, nt)
            assert bytes(area[len(want):]) == b"\xa5" * 64, (name, nt)  # nothing behind e264_expand_area_bytes
        n += 1
    assert n >= 10


def test_kernels_read_a_wire_packet_through_its_expansion(emu):
    """the four kernels' source on (wire packet + expansion buffer) against the same source on the version-4 packet: identical pictures and parameters"""
    pe, ie = emu
    for seed, w, h, kw in ((21, 20, 6, dict(p_skip=0.9, num_refs=2, residual_prob=0.3)), (22, 7, 5, dict(p_skip=0.7, intra_in_inter=0.2))):
        g = synth.StreamSynth(w, h, seed=seed, **kw)
        nb = P.frame_bytes(w, h)
        rng = np.random.default_rng(seed)
        dpb = [rng.integers(0, 256, nb + 64, dtype=np.uint8) for _ in range(6)] + [None] * 26
        for ft in "IPPB":
            pkt = bytes(g.next_frame(ft))
            wire = backend.packet_compact(pkt)
            if wire[4] != 5:
                Oracle().decode_frame(pkt, dpb, 3)
                continue
            res = []
            for form in (pkt, wire):
                mine = [None if b is None else b.copy() for b in dpb]
                scratch = np.full(146 * w * h + 64, 0x5A, np.uint8)
                prm = np.zeros(144 * w * h + 64, np.uint8)
                area = None
                if form is wire:
                    back = backend.packet_expand(wire)
                    hb = np.frombuffer(back, P.FRAME_HDR, 1)[0]
                    area = (C.c_uint8 * (int(hb["payload_off"]) - int(hb["mbs_off"])))()
                    pe.e264emu_expand(wire, area, 256)
                pe.e264emu_set_expand(area)
                ie.e264emu_set_expand(area)
                arr = _dpb_array(mine)
                assert pe.e264emu_pred_frame2(form, arr, scratch.ctypes.data) == 0
                assert ie.e264emu_intra_frame2(form, arr, scratch.ctypes.data, 1) == 0
                assert pe.e264emu_dbkparam_frame2(form, prm.ctypes.data, None) == 0
                d = int(P.Packet(pkt).hdr["dst_slot"])
                res.append((mine[d][:nb].copy(), prm[:144 * w * h].copy()))
            pe.e264emu_set_expand(None)
            ie.e264emu_set_expand(None)
            assert np.array_equal(res[0][0], res[1][0])
            assert np.array_equal(res[0][1], res[1][1])
            Oracle().decode_frame(pkt, dpb, 3)  # the next picture's references


def test_damaged_wire_packets_are_refused_or_sound():
    """bytes of the table, the entries and the header damaged at random: e264hip_packet_check either refuses the packet or its expansion is a sound
    version-4 packet (what the device unfolds is what was vetted)"""
    rng = np.random.default_rng(5)
    cases = _wire_cases()
    refused = accepted = 0
    for name, wire in cases:
        h = np.frombuffer(wire, P.FRAME_HDR, 1)[0]
        lo, hi = 0, int(h["payload_off"])
        for _ in range(150):
            buf = bytearray(wire)
            for _ in range(int(rng.integers(1, 4))):
                i = int(rng.integers(lo, hi))
                buf[i] = int(rng.integers(0, 256)) if rng.random() < 0.6 else buf[i] ^ (1 << int(rng.integers(0, 8)))
            buf = bytes(buf)
            if backend.packet_check(buf) != 0:
                refused += 1
                continue
            accepted += 1
            assert backend.packet_check(backend.packet_expand(buf)) == 0
    assert refused > 100 and accepted > 100
    # truncated / mislabelled
    name, wire = cases[0]
    assert backend.packet_check(wire[:-8]) != 0
    assert backend.packet_check(wire[:100]) != 0
    v4 = bytearray(wire)
    v4[4] = 4
    assert backend.packet_check(bytes(v4)) != 0
    with pytest.raises(backend.BackendError):
        backend.packet_compact(wire)  # already folded
    with pytest.raises(backend.BackendError):
        backend.packet_expand(synth_packets()[0][1])  # not a wire packet


@needs_front
def test_front_end_folds_pictures_that_leave_in_several_packets():
    """damaged streams (a slice cut short and sent again: the picture goes out in several packets whose records are edited in the packet -- E264_MBF_DONE, flags --
    before they are folded): packet by packet the fold of what the front end sends with the option off"""
    from tests import damage
    n5 = multi = 0
    cases = [damage.truncated_then_resent(*c) for c in damage.RESENT[:10]] + [damage.two_truncated_then_resent(*c) for c in damage.RESENT2[:4]]
    for data in cases:
        plain = [bytes(p) for p in front.capture_packets(data)[0]]
        wire = [bytes(p) for p in front.capture_packets(data, compact=True)[0]]
        assert len(plain) == len(wire)
        seen = {}
        for a, b in zip(plain, wire):
            assert backend.packet_check(b) == 0
            fid = int(P.Packet(a).hdr["frame_id"])
            seen[fid] = seen.get(fid, 0) + 1
            if b[4] == 5:
                assert b == backend.packet_compact(a)
                n5 += 1
            else:
                assert a == b
        multi += sum(1 for v in seen.values() if v > 1)
    front.capture_packets(b"", compact=False)
    assert n5 > 20 and multi > 5


@needs_front
def test_expand_kernel_source_on_large_pictures(emu):
    """1080p packets of an encoder-shaped stream (P and B pictures, 120 x 68 macroblocks: four bitmap words per row) and a synthetic 4096-wide picture (eight words per
    row): the kernel's source against the host expansion, and the fold's own round trip"""
    pe, _ = emu
    cases = []
    for name in ("nat1080_ipp30.264", "cabac_nat1080_ibbp30.264"):
        pk = front.capture_packets(open(os.path.join(STREAMS, name), "rb").read())[0]
        cases += [(name, backend.packet_compact(bytes(p))) for p in pk[1:8:3]]
    g = synth.StreamSynth(256, 3, seed=77, p_skip=0.8, num_refs=2)
    cases += [("wide", backend.packet_compact(bytes(g.next_frame(ft)))) for ft in "IPB"]
    n = 0
    for name, wire in cases:
        if wire[4] != 5:
            continue
        back = backend.packet_expand(wire)
        assert backend.packet_check(wire) == 0 and backend.packet_check(back) == 0 and backend.packet_compact(back) == wire, name
        h = np.frombuffer(back, P.FRAME_HDR, 1)[0]
        want = back[int(h["mbs_off"]):int(h["payload_off"])]
        area = (C.c_uint8 * (len(want) + 64))()
        C.memset(area, 0xA5, len(want) + 64)
        pe.e264emu_expand(wire, area, 256 * 3)
        assert bytes(area[:len(want)]) == want and bytes(area[len(want):]) == b"\xa5" * 64, name
        n += 1
    assert n >= 7
