"""The multi-stream driver's host side WITHOUT a GPU (edge264_amd/driver/e264_multi.cpp --parse-only: decoders advanced by a pool of
threads, any thread any decoder, packets collected per round but submitted nowhere): the packets it produces describe the same pictures as
the packets the capture sink produces for the same streams one decoder at a time (same sizes, macroblock counts, payloads; the DPB slot a
picture lands in depends on when the caller takes frames out, so the bytes may differ in the slot fields), every one passes the back end's
host-only validation, and nothing depends on the number of threads.  (On the GPU: tests/test_multi_stream.py.)"""
import collections
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STREAMS = os.path.join(HERE, "golden", "streams")
EXE = os.path.join(ROOT, "edge264_amd", "e264_multi")
FRONT = os.path.join(ROOT, "edge264_amd", "libedge264_hipfront.so")
HIP = os.path.join(ROOT, "edge264_amd", "libedge264_hip.so")
NAMES = ["ipb_spatial", "cabac_ipp", "mvc_ipp", "slices_deblock_idc", "cabac_t8x8_scaling", "i_4x4_16x16_pcm"]


def shape(p):
    """what a packet says about its picture, without the DPB slots"""
    from edge264_amd import packet as P
    h = P.Packet(p).hdr
    return tuple(int(h[k]) for k in ("total_bytes", "width_mbs", "height_mbs", "n_slices", "n_coded_mbs", "n_inter_mbs", "payload_bytes"))


def split_packets(data):
    out, off = [], 0
    while off < len(data):
        n = int.from_bytes(data[off + 8:off + 12], "little")  # E264FrameHdr.total_bytes
        out.append(data[off:off + n])
        off += n
    assert off == len(data)
    return out


@pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(FRONT) and os.path.exists(HIP)), reason="libraries not built")
def test_parse_only_driver_produces_the_capture_sinks_packets(tmp_path):
    from edge264_amd import backend, front
    files = [os.path.join(STREAMS, n + ".264") for n in NAMES]
    want = collections.Counter()
    frames = 0
    for f in files:
        pk, nf, _ = front.capture_packets(open(f, "rb").read())
        frames += nf
        for p in pk:
            assert backend.packet_check(p) == 0
            want[shape(p)] += 1
    repeat = 3
    stats = {}
    for threads in (1, 4):
        dump = tmp_path / f"p{threads}.e264"
        extra = ["--stay", "--ahead", "5"] if threads == 4 else []  # (round 5: a thread parses several pictures of one decoder in a row)
        out = subprocess.run([EXE, "--front", FRONT, "--hip", HIP, "--threads", str(threads), "--repeat", str(repeat), "--parse-only", "--dump-packets", str(dump)] + extra + files,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        st = stats[threads] = json.loads(out.stdout.strip().splitlines()[-1])
        assert st["streams"] == len(NAMES) * repeat and st["frames"] == repeat * frames
        pkts = split_packets(dump.read_bytes())
        for p in pkts:
            assert backend.packet_check(p) == 0
        got = collections.Counter(shape(p) for p in pkts)
        assert sum(got.values()) == st["packets"] == repeat * sum(want.values())
        assert got == collections.Counter({k: v * repeat for k, v in want.items()}), "the driver's packets do not describe the capture sink's pictures"
    assert stats[1]["packets"] == stats[4]["packets"]


@pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(FRONT)), reason="libraries not built")
def test_a_decoder_whose_picture_never_completes_does_not_hold_the_driver(tmp_path):
    """A slice that fails and never comes again leaves its picture incomplete: edge264_decode_NAL answers ENOBUFS and edge264_get_frame has nothing
    to hand out (the unmodified reference behaves the same: tests/test_frontend_concealment.py::test_failed_slice_never_resent).  Until round 5 the
    driver offered the same NAL again for ever.  Now it takes the API's way out (edge264_flush), every decoder ends, the other decoders of the run
    are not disturbed and each play of a damaged stream hands out what the reference hands out for it."""
    from tests import damage
    from edge264_amd import front
    want = {("ipp_partitions", 3, 0.5): 3, ("cabac_t8x8_slices", 11, 0.5): 3, ("nat_small_aq_slices_ipp8", 22, 0.5): 7, ("cabac_nat_small_aq_slices_ibbp10", 28, 0.4): 8}
    files = []
    for (n, w, k) in want:
        p = tmp_path / f"lost-{n}.264"
        p.write_bytes(damage.truncated_only(n, w, k))
        files.append(str(p))
    intact = [os.path.join(STREAMS, n + ".264") for n in ("cabac_ipp", "nat_small_rect_ipp8")]
    n_intact = sum(front.capture_packets(open(f, "rb").read())[1] for f in intact)
    for threads in (1, 3):
        out = subprocess.run([EXE, "--front", FRONT, "--hip", "/nonexistent/libedge264_hip.so", "--threads", str(threads), "--repeat", "2", "--parse-only", "--stay", "--ahead", "5"]
                             + files + intact, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        st = json.loads(out.stdout.strip().splitlines()[-1])
        assert st["frames"] == 2 * (sum(want.values()) + n_intact), st
        assert st["stuck_decoders_flushed"] == 2 * len(want) and st["stuck_decoders_given_up"] == 0, st
