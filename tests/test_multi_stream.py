"""Batched multi-stream front end (edge264_amd/driver/e264_multi.cpp) on the MI355X: many decoder instances
(reference parsers + our emitters, sink 2) advanced round-robin, the finished frames of a round submitted as ONE
batch; every output frame must equal the unmodified reference decoder's (tests/golden/streams/reference_md5.json)."""
import hashlib
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

pytestmark = pytest.mark.gpu


def frame_md5s(path, w_mbs, h_mbs, views=1):
    n = w_mbs * 16 * h_mbs * 16 * 3 // 2 * views
    data = open(path, "rb").read()
    assert len(data) % n == 0
    return [hashlib.md5(data[i:i + n]).hexdigest() for i in range(0, len(data), n)]


@pytest.mark.parametrize("names,repeat", [
    (["ipb_spatial", "t8x8_scaling", "slices_deblock_idc", "weighted_explicit", "one_mb", "tall_narrow", "i_4x4_16x16_pcm"], 3),
    (["hd1080_ippb", "cabac_hd1080_ipp"], 4),
    (["cabac_i", "cabac_ipp", "cabac_t8x8_scaling", "cabac_slices_deblock_idc", "cabac_weighted", "cabac_big_levels",
      "cabac_ipb_spatial", "cabac_ipb_temporal_implicit", "cabac_weighted_b"], 2),
    (["mvc_ipp", "mvc_cabac_ipb", "cabac_t8x8_slices", "mvc_ipb", "ipp_partitions"], 2),   # two views: two packets per access unit
])
def test_multi_stream_driver(tmp_path, names, repeat):
    for p in (EXE, FRONT, HIP):
        assert os.path.exists(p), f"{p} missing (built by __graft_entry__.build())"
    with open(os.path.join(STREAMS, "reference_md5.json")) as f:
        sums = json.load(f)
    files = [os.path.join(STREAMS, n + ".264") for n in names]
    dump = tmp_path / "packets.e264"
    out = subprocess.run([EXE, "--front", FRONT, "--hip", HIP, "--repeat", str(repeat), "--threads", "3", "--out", str(tmp_path),
                          "--dump-packets", str(dump)] + files, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    assert stats["streams"] == len(names) * repeat
    assert stats["frames"] == repeat * sum(len(sums[n]["md5"]) for n in names)
    assert stats["avg_batch"] > 1.2  # frames of different streams really share launches (batches are paced by the device: how many depends on timing)
    k = 0
    for _ in range(repeat):
        for n in names:
            got = frame_md5s(tmp_path / f"s{k}.yuv", sums[n]["width_mbs"], sums[n]["height_mbs"], sums[n]["views"])
            assert got == sums[n]["md5"], f"stream {k} ({n})"
            k += 1
    # decode-to-device mode: nothing is copied back, same frame count (and the other packet path: pageable packets, validated and
    # copied to staging memory by the back end, instead of page-locked ones submitted in place)
    out2 = subprocess.run([EXE, "--front", FRONT, "--hip", HIP, "--repeat", str(repeat), "--threads", "2", "--no-download", "--pageable"] + files,
                          capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    assert json.loads(out2.stdout.strip().splitlines()[-1])["frames"] == stats["frames"]
    # the packet dump is the capture format: self-describing records
    from edge264_amd import packet as P
    data = dump.read_bytes()
    off, npk = 0, 0
    while off < len(data):
        pk = P.Packet(data[off:off + int.from_bytes(data[off + 8:off + 12], "little")])
        off += int(pk.hdr["total_bytes"])
        npk += 1
    assert off == len(data) and npk == stats["packets"]
    print(stats)


def test_multi_stream_driver_devices(tmp_path):
    """--devices: decoder k lives on devices[k mod N], one submitter thread and one batch stream per entry.  One GPU here, so
    it is named twice: two submitters feed the same device object concurrently; every frame still equals the reference's."""
    with open(os.path.join(STREAMS, "reference_md5.json")) as f:
        sums = json.load(f)
    names = ["ipb_spatial", "cabac_ipp", "t8x8_scaling", "cabac_weighted_b", "slices_deblock_idc"]
    files = [os.path.join(STREAMS, n + ".264") for n in names]
    out = subprocess.run([EXE, "--front", FRONT, "--hip", HIP, "--devices", "0,0", "--repeat", "3", "--threads", "4", "--out", str(tmp_path)] + files,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    assert stats["devices"] == 2 and stats["streams"] == 15
    assert stats["frames"] == 3 * sum(len(sums[n]["md5"]) for n in names)
    k = 0
    for _ in range(3):
        for n in names:
            assert frame_md5s(tmp_path / f"s{k}.yuv", sums[n]["width_mbs"], sums[n]["height_mbs"], sums[n]["views"]) == sums[n]["md5"], f"stream {k} ({n})"
            k += 1


def test_multi_stream_driver_survives_a_lost_slice(tmp_path):
    """Decoders whose last picture never completes (a slice failed and never came again: edge264_decode_NAL answers ENOBUFS, nothing can be handed
    out) beside intact ones, on the device: the driver gets them out with edge264_flush instead of offering the same NAL for ever, every frame
    handed out equals the reference's (tests/golden/make_damage_md5.py), the intact decoders of the same batches are untouched."""
    from tests import damage
    with open(os.path.join(STREAMS, "damage_md5.json")) as f:
        dsums = json.load(f)
    with open(os.path.join(STREAMS, "reference_md5.json")) as f:
        sums = json.load(f)
    lost = [("ipp_partitions", 3, 0.5), ("cabac_t8x8_slices", 11, 0.5), ("nat_small_aq_slices_ipp8", 22, 0.5), ("cabac_nat_small_aq_slices_ibbp10", 28, 0.4)]
    intact = ["cabac_ipp", "nat_small_rect_ipp8", "slices_deblock_idc"]
    files = []
    for (n, w, k) in lost:
        p = tmp_path / f"lost-{n}.264"
        p.write_bytes(damage.truncated_only(n, w, k))
        files.append(str(p))
    files += [os.path.join(STREAMS, n + ".264") for n in intact]
    repeat = 2
    out = subprocess.run([EXE, "--front", FRONT, "--hip", HIP, "--repeat", str(repeat), "--threads", "3", "--out", str(tmp_path)] + files,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    st = json.loads(out.stdout.strip().splitlines()[-1])
    assert st["stuck_decoders_flushed"] == repeat * len(lost) and st["stuck_decoders_given_up"] == 0, st
    k = 0
    for _ in range(repeat):
        for (n, w, kp) in lost:
            assert frame_md5s(tmp_path / f"s{k}.yuv", sums[n]["width_mbs"], sums[n]["height_mbs"]) == dsums[f"lost-{n}-{w}-{kp}"]["md5"], f"stream {k} (lost {n})"
            k += 1
        for n in intact:
            assert frame_md5s(tmp_path / f"s{k}.yuv", sums[n]["width_mbs"], sums[n]["height_mbs"]) == sums[n]["md5"], f"stream {k} ({n})"
            k += 1
