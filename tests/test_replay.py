"""Capture files: packets of several decoders interleaved in one file, read back and replayed.
CPU: the file format round-trips and the batching never puts a stream twice in one submission.
GPU: the replayer's per-picture md5s equal the oracle's on the same capture (real bitstreams through the reference's
front end: real motion / partition statistics)."""
import glob
import hashlib
import os

import numpy as np
import pytest

from edge264_amd import packet as P, replay

HERE = os.path.dirname(os.path.abspath(__file__))
STREAMS = os.path.join(HERE, "golden", "streams")
REF = os.path.join(os.path.dirname(HERE), "oracle", "_ref")
needs_front = pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(REF), "..", "edge264_amd", "libedge264_hipfront.so")),
                                 reason="edge264_amd/libedge264_hipfront.so is built from /root/reference (make -C edge264_amd/frontend)")
NAMES = ["ipb_spatial", "cabac_ipb_temporal_implicit", "weighted_explicit", "t8x8_scaling", "mvc_ipb"]


def make_capture(tmp_path, oracle):
    """The fixtures' packets (reference front end, capture sink), interleaved round-robin like e264_multi writes them."""
    from oracle.pyoracle import HipFront
    per_stream = []
    for n in NAMES:
        _, _, pk = HipFront().decode_capture(open(os.path.join(STREAMS, n + ".264"), "rb").read(), oracle)
        per_stream.append(pk)
    packets, sids = [], []
    for i in range(max(len(p) for p in per_stream)):
        for sid, pk in enumerate(per_stream):
            if i < len(pk):
                packets.append(pk[i])
                sids.append(sid)
    path = str(tmp_path / "capture.e264")
    replay.Capture.write(path, packets, sids)
    return path, per_stream


def oracle_md5s(oracle, packets):
    dpb, out = [None] * 32, []
    for pkt in packets:
        h = P.Packet(pkt).hdr
        nb = int(h["plane_size_Y"]) + int(h["plane_size_C"])
        for s in range(32):
            if dpb[s] is None and (s == int(h["dst_slot"]) or int(h["ref_slots"]) >> s & 1):
                dpb[s] = np.zeros(nb + 64, np.uint8)
        oracle.decode_frame(pkt, dpb, 3)
        out.append(hashlib.md5(dpb[int(h["dst_slot"])][:nb].tobytes()).hexdigest())
    return out


@needs_front
def test_capture_file_round_trip(tmp_path, oracle):
    path, per_stream = make_capture(tmp_path, oracle)
    cap = replay.Capture.load(path)
    assert len(cap.packets) == sum(len(p) for p in per_stream)
    for sid, pk in enumerate(per_stream):
        got = cap.of_stream(sid)
        assert len(got) == len(pk)
        for a, b in zip(got, pk):  # identical apart from the stream tag
            assert a[:76] == b[:76] and a[80:] == b[80:]
    for run in replay.batches(cap):
        ids = [sid for sid, _ in run]
        assert len(ids) == len(set(ids)) and len(ids) >= 1
    with pytest.raises(ValueError):
        replay.Capture(open(path, "rb").read()[:-5])


@needs_front
@pytest.mark.gpu
def test_replay_matches_oracle_on_the_gpu(tmp_path, oracle):
    from edge264_amd import backend
    path, per_stream = make_capture(tmp_path, oracle)
    dev = backend.Device(0)
    try:
        got = replay.replay(replay.Capture.load(path), dev)
    finally:
        dev.close()
    assert sorted(got) == list(range(len(NAMES)))
    for sid, pk in enumerate(per_stream):
        assert got[sid] == oracle_md5s(oracle, pk), NAMES[sid]
