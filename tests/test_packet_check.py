"""e264hip_packet_check (include/edge264_hip.h): the host-side validation every host-packet entry point runs
before a packet may reach the device.  Host-only, so it is tested on the CPU: packets from the synthesiser and
from the reference front end (all fixtures) pass; each corruption a kernel would dereference is EINVAL, the
reference's error convention (positive errno, README.md:188-209), never a fault.
"""
import errno
import glob
import os

import numpy as np
import pytest

from edge264_amd import backend, packet as P, synth

HERE = os.path.dirname(os.path.abspath(__file__))
STREAMS = os.path.join(HERE, "golden", "streams")
REF = os.path.join(os.path.dirname(HERE), "oracle", "_ref")


@pytest.fixture(scope="module")
def gop():
    return synth.StreamSynth(6, 4, seed=11, num_refs=2).gop("IPB")


def test_synth_packets_pass(gop):
    for raw in gop:
        assert backend.packet_check(raw) == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(REF), "..", "edge264_amd", "libedge264_hipfront.so")), reason="front end not built")
def test_front_end_packets_pass(oracle):
    from oracle.pyoracle import HipFront
    n = 0
    for path in sorted(glob.glob(os.path.join(STREAMS, "*.264"))):
        if "1080" in path:
            continue
        _, _, packets = HipFront().decode_capture(open(path, "rb").read(), oracle)
        for raw in packets:
            assert backend.packet_check(raw) == 0, os.path.basename(path)
            n += 1
    assert n > 40


def mutate(raw, fn):
    buf = bytearray(raw)
    pk = P.Packet(buf)   # numpy views over the bytearray are writable
    fn(pk, buf)
    return bytes(buf)


def first(pk, kind):
    return int(np.nonzero(pk.mbs["kind"] == kind)[0][0])


def hdr_set(field, value):
    def fn(pk, buf):
        h = np.frombuffer(buf, P.FRAME_HDR, 1)
        h[field] = value
    return fn


def mb_set(field, value, kind=None):
    def fn(pk, buf):
        mbs = np.frombuffer(buf, P.MB, len(pk.mbs), int(pk.hdr["mbs_off"]))
        a = 7 if kind is None else first(pk, kind)
        mbs[field][a] = value
    return fn


def motion_set(byte, value):
    """Overwrite one byte of the first inter macroblock's compact motion record (version 3: byte 0 = refPic and byte 1 = refIdx
    of its first predicted part)."""
    def fn(pk, buf):
        a = first(pk, P.MB_INTER)
        mbs = np.frombuffer(buf, P.MB, len(pk.mbs), int(pk.hdr["mbs_off"]))
        off = int(mbs["modes"][a][:4].view("<u4")[0])
        buf[int(pk.hdr["motion_off"]) + off + byte] = value & 255
    return fn


def directory_set(word, value):
    def fn(pk, buf):
        a = first(pk, P.MB_INTER)
        mbs = np.frombuffer(buf, P.MB, len(pk.mbs), int(pk.hdr["mbs_off"]))
        mbs["modes"][a][word * 4:word * 4 + 4] = np.frombuffer(np.uint32(value).tobytes(), np.uint8)
    return fn


CORRUPTIONS = [
    ("magic", 0, hdr_set("magic", 0x12345678)),
    ("version", 0, hdr_set("version", 99)),
    ("total_bytes", 0, hdr_set("total_bytes", 1 << 30)),
    ("dst_slot", 0, hdr_set("dst_slot", 32)),
    ("dst_slot_neg", 0, hdr_set("dst_slot", -1)),
    ("zero_width", 0, hdr_set("width_mbs", 0)),
    ("huge_frame", 0, hdr_set("height_mbs", 4000)),
    ("mbs_off", 0, hdr_set("mbs_off", 1 << 28)),
    ("payload_off", 0, hdr_set("payload_off", 1 << 28)),
    ("payload_bytes", 0, hdr_set("payload_bytes", 1 << 28)),
    ("motion_off", 1, hdr_set("motion_off", 1 << 28)),
    ("motion_missing", 1, hdr_set("motion_off", 0)),
    ("n_slices", 0, hdr_set("n_slices", 1 << 20)),
    ("no_slices", 0, hdr_set("n_slices", 0)),
    ("stride_small", 0, hdr_set("stride_Y", 16)),
    ("plane_small", 0, hdr_set("plane_size_Y", 64)),
    ("kind", 0, mb_set("kind", 9)),
    ("slice_index", 0, mb_set("slice", 5)),
    ("mb_payload_off", 0, mb_set("payload_off", 1 << 27)),
    ("mb_payload_align", 0, mb_set("payload_off", 4)),
    ("coded_overrun", 0, mb_set("payload_off", None)),     # filled in below: last 8 bytes of the payload
    ("t8x8_on_i16x16", 0, None),                            # filled in below
    ("absent_wild_slice", 0, None),                         # filled in below: ABSENT records are read by the parameter kernel too
    ("refPic", 1, motion_set(0, 40)),
    ("refPic_neg", 1, motion_set(0, -3)),
    ("refPic_none", 1, motion_set(0, -1)),                  # a part the directory announces must name a picture
    ("refIdx", 1, motion_set(1, 77)),
    ("mot_off_align", 1, directory_set(0, 2)),
    ("mot_off_overrun", 1, directory_set(0, 1 << 26)),
    ("mot_hdr_bits", 1, directory_set(1, 1 << 27)),
]


@pytest.mark.parametrize("name,frame,fn", CORRUPTIONS, ids=[c[0] for c in CORRUPTIONS])
def test_corruption_is_einval(gop, name, frame, fn):
    raw = gop[frame]
    if name == "coded_overrun":
        def fn(pk, buf):
            mbs = np.frombuffer(buf, P.MB, len(pk.mbs), int(pk.hdr["mbs_off"]))
            a = int(np.nonzero(mbs["coded"] != 0)[0][0])
            mbs["payload_off"][a] = (int(pk.hdr["payload_bytes"]) - 8) & ~7
    if name == "t8x8_on_i16x16":
        def fn(pk, buf):
            mbs = np.frombuffer(buf, P.MB, len(pk.mbs), int(pk.hdr["mbs_off"]))
            mbs["flags"][first(pk, P.MB_I16x16)] |= P.MBF_T8x8
    if name == "absent_wild_slice":
        def fn(pk, buf):
            mbs = np.frombuffer(buf, P.MB, len(pk.mbs), int(pk.hdr["mbs_off"]))
            mbs["kind"][3] = P.MB_ABSENT
            mbs["slice"][3] = 60000
    bad = mutate(raw, fn)
    assert bad != raw
    assert backend.packet_check(bad) == errno.EINVAL
    assert backend.last_error()


def test_truncated_packet(gop):
    for cut in (0, 16, 79, 80, len(gop[1]) // 2, len(gop[1]) - 1):
        assert backend.packet_check(gop[1][:cut]) == errno.EINVAL


def test_byte_fuzz_never_faults(gop):
    """Random byte flips in the header / record sections: any verdict is fine, a crash is not."""
    rng = np.random.default_rng(3)
    raw = gop[2]
    pk = P.Packet(raw)
    limit = int(pk.hdr["payload_off"])
    for _ in range(300):
        buf = bytearray(raw)
        for _ in range(int(rng.integers(1, 6))):
            buf[int(rng.integers(0, limit))] = int(rng.integers(0, 256))
        assert backend.packet_check(bytes(buf)) in (0, errno.EINVAL)
