"""Wire packets (version 5, include/edge264_compact.h) through every host-packet entry point of the C-ABI on the MI355X: e264_expand_kernel in
front of the four kernels.  Expectation: the pictures the version-4 packets give through the same entry point (whose parity with the oracle and the
reference is the business of the other -m gpu tests) and, for the front end with e264front_set_compact(1), the unmodified reference's md5s."""
import ctypes as C
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from edge264_amd import backend, front, packet as P, synth

HERE = os.path.dirname(os.path.abspath(__file__))
STREAMS = os.path.join(HERE, "golden", "streams")
FRONT = os.path.join(os.path.dirname(HERE), "edge264_amd", "libedge264_hipfront.so")
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(STREAMS, "*.264")))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    dev = backend.Device(0)
    yield dev
    dev.close()


def _gop(g, gop):
    return [bytes(g.next_frame(ft)) for ft in gop]


def synth_streams(n, w, h, gop, **kw):
    return [_gop(synth.StreamSynth(w, h, seed=100 + i, **kw), gop) for i in range(n)]


def run(device, per_stream, how):
    """per_stream: packet lists (one per stream, same length).  Picture i of every stream in one submission through entry point `how`.
    Returns [stream][picture] md5 of the destination slot."""
    n = len(per_stream)
    h0 = P.Packet(per_stream[0][0]).hdr
    sts = [backend.Stream(device, int(h0["width_mbs"]), int(h0["height_mbs"])) for _ in range(n)]
    allocated = [set() for _ in range(n)]
    out = [[] for _ in range(n)]
    pinned = []
    try:
        for i in range(len(per_stream[0])):
            pkts = [ps[i] for ps in per_stream]
            for st, al, pkt in zip(sts, allocated, pkts):
                h = P.Packet(pkt).hdr
                st.frame_bytes = int(h["plane_size_Y"]) + int(h["plane_size_C"])
                for s in range(P.MAX_SLOTS):
                    if (s == int(h["dst_slot"]) or int(h["ref_slots"]) >> s & 1) and s not in al:
                        st.alloc(s)
                        st.fill(s, 0)
                        al.add(s)
            if how == "single":
                for st, pkt in zip(sts, pkts):
                    st.submit(pkt)
            elif how == "host":
                device.submit_batch_host(sts, pkts)
            elif how in ("pinned", "pinned_untrusted"):
                ptrs = [device.pinned_copy(p) for p in pkts]
                pinned += ptrs
                device.submit_pinned_prepared(device.prepare_pinned_batch(sts, ptrs, [len(p) for p in pkts]), trusted=how == "pinned")
            elif how == "resident":
                dps = [device.upload_packet(p) for p in pkts]
                device.submit_batch(sts, dps)
                for dp in dps:
                    dp.free()
            for k, (st, pkt) in enumerate(zip(sts, pkts)):
                out[k].append(hashlib.md5(st.download(int(P.Packet(pkt).hdr["dst_slot"])).tobytes()).hexdigest())
    finally:
        device.sync()
        for p in pinned:
            device.pinned_free(p)
        for st in sts:
            st.close()
    return out


@pytest.mark.parametrize("how", ["single", "host", "pinned", "pinned_untrusted", "resident"])
def test_wire_packets_give_the_same_pictures(device, how):
    cases = [synth_streams(3, 20, 6, "IPPBP", p_skip=0.9, num_refs=2, residual_prob=0.3),
             synth_streams(40 if how != "single" else 2, 7, 5, "IPBP", p_skip=0.6, intra_in_inter=0.2),  # >= 32 pinned packets: the gathered transfer
             synth_streams(2, 65, 3, "IPP", p_skip=1.0, residual_prob=0.0)]
    folded = 0
    for per_stream in cases:
        wire = [[backend.packet_compact(p) for p in ps] for ps in per_stream]
        folded += sum(w[4] == 5 for ws in wire for w in ws)
        assert run(device, wire, how) == run(device, per_stream, how)
    assert folded > 20


def test_mixed_batches_and_growing_motion_sections(device):
    """one submission with version-4 and wire packets side by side; a stream whose later pictures need a larger expansion buffer"""
    a = synth_streams(4, 20, 6, "IPPPP", p_skip=0.9)
    mixed = [[backend.packet_compact(p) if (k + i) & 1 else p for i, p in enumerate(ps)] for k, ps in enumerate(a)]
    assert run(device, mixed, "host") == run(device, a, "host")
    g = synth.StreamSynth(20, 6, seed=9, p_skip=0.95)
    first = _gop(g, "IPP")
    g.p_skip = 0.3  # many partitioned macroblocks: longer motion records
    later = _gop(g, "PBPB")
    grow = [first + later]
    wire = [[backend.packet_compact(p) for p in grow[0]]]
    assert run(device, wire, "single") == run(device, grow, "single")


@pytest.fixture(scope="module")
def _hipfront_lib():
    if not os.path.exists(FRONT):
        pytest.fail(f"{FRONT} missing: it is built in the container by `make -C oracle ref` and travels with the snapshot")
    from oracle.pyoracle import HipFront
    h = HipFront()
    h.lib.e264front_set_compact.argtypes = [C.c_int]
    return h


@pytest.fixture
def hipfront(_hipfront_lib):
    """the front library on the device sink, folding (both switches are global to the library and other tests of this file move them: set per test)"""
    _hipfront_lib.lib.e264front_set_sink(0)
    _hipfront_lib.lib.e264front_set_compact(1)
    yield _hipfront_lib
    _hipfront_lib.lib.e264front_set_compact(0)


@pytest.mark.parametrize("name", NAMES)
def test_front_end_in_wire_form_matches_reference(name, hipfront):
    """every committed stream through edge264.h on the device sink with the front end folding its packets: the unmodified reference's frames and codes"""
    with open(os.path.join(STREAMS, "reference_md5.json")) as f:
        sums = json.load(f)
    data = open(os.path.join(STREAMS, name + ".264"), "rb").read()
    frames, codes = hipfront.decode(data)
    assert codes == sums[name]["nal_codes"]
    assert [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in frames] == sums[name]["md5"]


def test_captured_streams_as_wire_batches(device):
    """real streams' packets (front end, capture sink, folded by the front end itself), several streams per submission, against their version-4 form"""
    names = ["ipb_spatial", "cabac_ipb_temporal_implicit", "weighted_explicit", "nat_small_ipp8", "cabac_nat_small_ibbp10"]
    for nm in names:
        data = open(os.path.join(STREAMS, nm + ".264"), "rb").read()
        plain = [bytes(p) for p in front.capture_packets(data)[0]]
        wire = [bytes(p) for p in front.capture_packets(data, compact=True)[0]]
        front.capture_packets(b"", compact=False)
        assert any(w[4] == 5 for w in wire), nm
        assert run(device, [wire, wire], "host") == run(device, [plain, plain], "host"), nm


def test_multi_stream_driver_in_wire_form(tmp_path):
    """e264_multi (many decoders, parser threads, packets in page-locked memory submitted in place and trusted) with E264_FRONT_COMPACT=1: the reference's frames"""
    import subprocess
    root = os.path.dirname(HERE)
    exe, hip = os.path.join(root, "edge264_amd", "e264_multi"), os.path.join(root, "edge264_amd", "libedge264_hip.so")
    for p in (exe, FRONT, hip):
        assert os.path.exists(p), f"{p} missing (built by __graft_entry__.build())"
    with open(os.path.join(STREAMS, "reference_md5.json")) as f:
        sums = json.load(f)
    names = ["ipb_spatial", "cabac_ipb_temporal_implicit", "nat_small_ipp8", "cabac_nat_small_ibbp10", "weighted_explicit", "mvc_ipb", "cabac_nat_small_aq_slices_ibbp10"]
    files = [os.path.join(STREAMS, n + ".264") for n in names]
    out = subprocess.run([exe, "--front", FRONT, "--hip", hip, "--repeat", "3", "--threads", "3", "--out", str(tmp_path)] + files,
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, E264_FRONT_COMPACT="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    assert stats["frames"] == 3 * sum(len(sums[n]["md5"]) for n in names)
    k = 0
    for _ in range(3):
        for n in names:
            nby = sums[n]["width_mbs"] * 16 * sums[n]["height_mbs"] * 16 * 3 // 2 * sums[n]["views"]
            data = open(tmp_path / f"s{k}.yuv", "rb").read()
            assert [hashlib.md5(data[i:i + nby]).hexdigest() for i in range(0, len(data), nby)] == sums[n]["md5"], f"stream {k} ({n})"
            k += 1


def test_concealment_in_wire_form(hipfront):
    """tests/test_frontend_hip.py::test_concealment_on_the_gpu with the front end folding: pictures that leave in several packets (their records edited in the
    packet before the fold) through the device expansion -- the unmodified reference's frames"""
    from tests import damage
    with open(os.path.join(STREAMS, "damage_md5.json")) as f:
        sums = json.load(f)

    def md5s(frames):
        return [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in frames]
    for name, which, keep in damage.RESENT:
        frames, codes = hipfront.decode(damage.truncated_then_resent(name, which, keep))
        want = sums[f"{name}-{which}-{keep}"]
        assert codes == want["nal_codes"] and md5s(frames) == want["md5"], (name, which, keep)
    for name, which, ka, kb in damage.RESENT2:
        frames, codes = hipfront.decode(damage.two_truncated_then_resent(name, which, ka, kb))
        want = sums[f"{name}-{which}+{which + 1}-{ka}-{kb}"]
        assert codes == want["nal_codes"] and md5s(frames) == want["md5"], (name, which, ka, kb)
    for name in damage.DAMAGED_FILES:
        frames, codes = hipfront.decode(open(os.path.join(damage.DAMAGED_DIR, name + ".264"), "rb").read())
        want = sums[f"file-{name}"]
        assert codes == want["nal_codes"] and md5s(frames) == want["md5"], name
