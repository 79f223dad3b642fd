"""Submissions that mix I pictures with P / B pictures (streams whose GOPs are not in phase): the launcher starts the intra pass of the pictures without
prediction work on the second queue beside the others' parameter and prediction kernels (E264Fork.n_nopred, edge264_amd/csrc/e264_kernels.h).  Same pictures
as with the option off, as one stream at a time, and as the oracle's."""
import hashlib

import numpy as np
import pytest

from edge264_amd import backend, packet as P, synth
from tests.test_hip_wire import run, _gop

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    dev = backend.Device(0)
    yield dev
    dev.set_option("split_intra", 1)
    dev.close()


GOPS = ["IPPIPB", "IIPPPB", "IPIPBP", "IPPPIP", "IPBIPP", "IIIPIP", "IPPPPP"]


def streams(n, w, h, **kw):
    return [_gop(synth.StreamSynth(w, h, seed=300 + i, **kw), GOPS[i % len(GOPS)]) for i in range(n)]


@pytest.mark.parametrize("how", ["resident", "host", "pinned", "pinned_untrusted"])
def test_mixed_submissions(device, how, oracle):
    per_stream = streams(11, 20, 6, num_refs=2, intra_in_inter=0.2, t8x8=True)
    device.set_option("split_intra", 1)
    got = run(device, per_stream, how)
    device.set_option("split_planes", 0)  # the split-off pictures' intra pass on one workgroup instead of two (luma, chroma)
    assert got == run(device, per_stream, how)
    device.set_option("split_planes", 1)
    device.set_option("split_intra", 0)
    assert got == run(device, per_stream, how)
    device.set_option("split_intra", 1)
    assert got == [run(device, [ps], "single")[0] for ps in per_stream]
    # ... and the oracle's pictures for three of the streams
    for k in (0, 3, 5):
        nb = P.frame_bytes(20, 6)
        dpb = [np.zeros(nb + 64, np.uint8) for _ in range(8)] + [None] * 24
        for i, pkt in enumerate(per_stream[k]):
            oracle.decode_frame(pkt, dpb, 3)
            assert hashlib.md5(dpb[int(P.Packet(pkt).hdr["dst_slot"])][:nb].tobytes()).hexdigest() == got[k][i], (k, i)


@pytest.mark.parametrize("how", ["resident", "pinned"])
def test_mixed_submissions_with_pcm(device, how):
    """I pictures WITH I_PCM macroblocks have work for the prediction kernel and stay with the others; those without are split off -- both kinds in one batch"""
    per_stream = [_gop(synth.StreamSynth(9, 5, seed=500 + i, pcm_prob=0.3 if i & 1 else 0.0, intra_in_inter=0.2), GOPS[i % len(GOPS)]) for i in range(9)]
    device.set_option("split_intra", 1)
    got = run(device, per_stream, how)
    device.set_option("split_intra", 0)
    assert got == run(device, per_stream, how)
    device.set_option("split_intra", 1)
    assert got == [run(device, [ps], "single")[0] for ps in per_stream]


def test_mixed_submissions_on_two_lanes(device):
    """two lanes share the device's second queue: their split submissions interleave"""
    per_stream = streams(8, 9, 5)
    want = run(device, per_stream, "resident")
    n = len(per_stream)
    sts = [backend.Stream(device, 9, 5) for _ in range(n)]
    try:
        for k, st in enumerate(sts):
            st.bind_lane(k & 1)
            for s in range(8):
                st.alloc(s)
                st.fill(s, 0)
        got = [[] for _ in range(n)]
        for i in range(len(per_stream[0])):
            for lane in (0, 1):
                idx = [k for k in range(n) if (k & 1) == lane]
                dps = [device.upload_packet(per_stream[k][i]) for k in idx]
                device.submit_batch([sts[k] for k in idx], dps)
                for dp in dps:
                    dp.free()
            for k in range(n):
                got[k].append(hashlib.md5(sts[k].download(int(P.Packet(per_stream[k][i]).hdr["dst_slot"])).tobytes()).hexdigest())
        assert got == want
    finally:
        device.sync()
        for st in sts:
            st.close()
