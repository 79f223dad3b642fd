"""End-to-end drop-in test on the MI355X: the edge264.h API (edge264_alloc / decode_NAL / get_frame)
served by the reference's front end + our emitters + libedge264_hip.so (HIP sink), on the committed
Annex-B fixtures; frames must equal what the unmodified reference decoder produced
(tests/golden/streams/reference_md5.json, written by tests/golden/make_streams.py)."""
import glob
import hashlib
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
STREAMS = os.path.join(HERE, "golden", "streams")
FRONT = os.path.join(os.path.dirname(HERE), "edge264_amd", "libedge264_hipfront.so")
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(STREAMS, "*.264")))

pytestmark = pytest.mark.gpu


def md5s(frames):
    return [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in frames]


@pytest.fixture(scope="module")
def front():
    if not os.path.exists(FRONT):
        pytest.fail(f"{FRONT} missing: it is built in the container by `make -C oracle ref` and travels with the snapshot")
    from oracle.pyoracle import HipFront
    h = HipFront()
    h.lib.e264front_set_sink(0)
    return h


@pytest.mark.parametrize("name", NAMES)
def test_hip_sink_matches_reference(name, front):
    with open(os.path.join(STREAMS, "reference_md5.json")) as f:
        sums = json.load(f)
    data = open(os.path.join(STREAMS, name + ".264"), "rb").read()
    frames, codes = front.decode(data)
    assert codes == sums[name]["nal_codes"]
    assert md5s(frames) == sums[name]["md5"]


API = os.path.join(HERE, "golden", "api")
API_NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(API, "*.264")))


@pytest.mark.parametrize("name", API_NAMES)
def test_reference_api_fixtures_on_the_hip_sink(name, front):
    """The reference's own API-behaviour fixtures (/root/reference/tests/*.264, copied as data into tests/golden/api: parameter sets missing or
    changing, frame finishing, POC order, nal_ref_idc 0, supported / unsupported NAL types, zero cropping) through the product on the GPU: the same
    return code for every edge264_decode_NAL call and the same frames as the unmodified reference (tests/golden/make_api_md5.py)."""
    with open(os.path.join(API, "api_md5.json")) as f:
        sums = json.load(f)
    data = open(os.path.join(API, name + ".264"), "rb").read()
    frames, codes = front.decode(data)
    assert codes == sums[name]["nal_codes"]
    assert md5s(frames) == sums[name]["md5"]


def test_two_decoders_interleaved(front):
    """Two decoder instances share the device object (one E264Stream each)."""
    import ctypes as C
    import numpy as np
    from oracle.pyoracle import Edge264Frame
    with open(os.path.join(STREAMS, "reference_md5.json")) as f:
        sums = json.load(f)
    L = front.lib
    names = ["ipb_spatial", "t8x8_scaling"]
    st = []
    for n in names:
        data = open(os.path.join(STREAMS, n + ".264"), "rb").read()
        buf = np.frombuffer(data + b"\0" * 64, np.uint8).copy()
        base = buf.ctypes.data
        dec = C.c_void_p(L.edge264_alloc(0, None, None, 0, None, None, None))
        assert dec
        nal = L.edge264_find_start_code(base, base + len(data), 0) + 3
        st.append(dict(buf=buf, end=base + len(data), dec=dec, nal=nal, frames=[], done=False))
    out = Edge264Frame()
    while not all(s["done"] for s in st):
        for s in st:
            if s["done"]:
                continue
            nxt = L.edge264_find_start_code(s["nal"], s["end"], 0) if s["nal"] < s["end"] else s["end"]
            res = L.edge264_decode_NAL(s["dec"], s["nal"], nxt, None, None)
            while L.edge264_get_frame(s["dec"], C.byref(out), 0) == 0:
                s["frames"].append(front._copy_frame(out))
            if res == 105:
                continue
            if res == 61 or s["nal"] >= s["end"]:
                s["done"] = True
            s["nal"] = min(nxt + 3, s["end"])
    for n, s in zip(names, st):
        L.edge264_free(C.byref(s["dec"]))
        assert md5s(s["frames"]) == sums[n]["md5"]


def test_caller_allocators_receive_the_frames(front):
    """edge264_alloc with alloc_cb / free_cb (edge264.h:42-43) on the HIP sink: Edge264Frame.samples point into the CALLER's
    blocks, edge264_get_frame fills them from HBM, frames equal the reference's; n_threads > 0 at the same time (a hint)."""
    import ctypes as C
    import numpy as np
    from oracle.pyoracle import CallerAllocator, Edge264Frame
    with open(os.path.join(STREAMS, "reference_md5.json")) as f:
        sums = json.load(f)
    L = front.lib
    for name in ("ipb_spatial", "i_4x4_16x16_pcm"):
        data = open(os.path.join(STREAMS, name + ".264"), "rb").read()
        al = CallerAllocator()
        buf = np.frombuffer(data + b"\0" * 64, np.uint8).copy()
        base, end = buf.ctypes.data, buf.ctypes.data + len(data)
        dec = C.c_void_p(L.edge264_alloc(3, None, None, 0, *al.args()))
        assert dec
        frames, out = [], Edge264Frame()
        nal = L.edge264_find_start_code(base, end, 0) + 3
        while True:
            nxt = L.edge264_find_start_code(nal, end, 0) if nal < end else end
            res = L.edge264_decode_NAL(dec, nal, nxt, None, None)
            while L.edge264_get_frame(dec, C.byref(out), 0) == 0:
                assert al.owns(out.samples[0]) and al.owns(out.samples[1]) and al.owns(out.samples[2])
                frames.append(front._copy_frame(out))
            if res == 105:
                continue
            if res == 61 or nal >= end:
                break
            nal = min(nxt + 3, end)
        L.edge264_free(C.byref(dec))
        assert md5s(frames) == sums[name]["md5"]
        assert al.allocs == al.frees and not al.live


def test_concealment_on_the_gpu(front):
    """tests/test_frontend_concealment.py on the device: a slice cut short, then sent again -- the failed attempt, its
    deblocking, the P_Skip / B_Skip concealment and the second decode reach the kernels as a sequence of packets per picture
    (E264_MBF_DONE); every frame must equal the unmodified reference decoder's (md5s computed in the container by
    tests/golden/make_damage_md5.py from the reference itself: /root/reference does not travel)."""
    from tests import damage
    with open(os.path.join(STREAMS, "damage_md5.json")) as f:
        sums = json.load(f)
    for name, which, keep in damage.RESENT:
        frames, codes = front.decode(damage.truncated_then_resent(name, which, keep))
        want = sums[f"{name}-{which}-{keep}"]
        assert codes == want["nal_codes"], (name, which, keep)
        assert md5s(frames) == want["md5"], (name, which, keep)
    # two failed slices inside one picture before either arrives again
    for name, which, ka, kb in damage.RESENT2:
        frames, codes = front.decode(damage.two_truncated_then_resent(name, which, ka, kb))
        want = sums[f"{name}-{which}+{which + 1}-{ka}-{kb}"]
        assert codes == want["nal_codes"], (name, which, ka, kb)
        assert md5s(frames) == want["md5"], (name, which, ka, kb)
    # cases of tools/damage_sweep.py kept as files: a stray copy of a slice whose picture has already gone out
    for name in damage.DAMAGED_FILES:
        frames, codes = front.decode(open(os.path.join(damage.DAMAGED_DIR, name + ".264"), "rb").read())
        want = sums[f"file-{name}"]
        assert codes == want["nal_codes"] and md5s(frames) == want["md5"], name
    # a slice lost for good: the stream ends stuck (ENOBUFS with nothing to hand out), everything before is the reference's
    for name, which, keep in damage.LOST:
        frames, codes = front.decode(damage.truncated_only(name, which, keep))
        want = sums[f"lost-{name}-{which}-{keep}"]
        assert codes == want["nal_codes"], (name, which, keep)
        assert md5s(frames) == want["md5"], (name, which, keep)


def test_decode_to_device_without_readback(front):
    """Decode-to-device: with e264front_set_download(0) edge264_get_frame copies nothing back; the picture is
    read where it lies in HBM through e264front_device_samples (the device address behind an Edge264Frame plane pointer, same
    strides) -- here with a plain hipMemcpy -- and equals the reference's."""
    import ctypes as C
    import errno
    import numpy as np
    from oracle.pyoracle import Edge264Frame
    with open(os.path.join(STREAMS, "reference_md5.json")) as f:
        sums = json.load(f)
    L = front.lib
    L.e264front_device_samples.restype = C.c_void_p
    L.e264front_device_samples.argtypes = [C.c_void_p, C.c_void_p]
    hipl = C.CDLL("libamdhip64.so")
    hipl.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    name = "ipb_spatial"
    data = open(os.path.join(STREAMS, name + ".264"), "rb").read()
    buf = np.frombuffer(data + b"\0" * 64, np.uint8).copy()
    base, end = buf.ctypes.data, buf.ctypes.data + len(data)
    L.e264front_set_download(0)
    try:
        dec = C.c_void_p(L.edge264_alloc(0, None, None, 0, None, None, None))
        assert dec
        out = Edge264Frame()
        frames = []

        def drain():
            while L.edge264_get_frame(dec, C.byref(out), 0) == 0:
                planes = []
                for i, (w, h, stride) in enumerate(((out.width_Y, out.height_Y, out.stride_Y), (out.width_C, out.height_C, out.stride_C), (out.width_C, out.height_C, out.stride_C))):
                    dptr = L.e264front_device_samples(dec, out.samples[i])
                    assert dptr, "no device address for a frame the decoder handed out"
                    host = np.zeros((h, stride), np.uint8)
                    assert hipl.hipMemcpy(host.ctypes.data, dptr, host.nbytes - (stride - w), 2) == 0  # device -> host (the last row ends at its last sample)
                    planes.append(host[:, :w].copy())
                frames.append(tuple(planes))
        nal = L.edge264_find_start_code(base, end, 0) + 3
        while True:
            nxt = L.edge264_find_start_code(nal, end, 0) if nal < end else end
            res = L.edge264_decode_NAL(dec, nal, nxt, None, None)
            n0 = len(frames)
            drain()
            if res == errno.ENOBUFS:
                assert len(frames) > n0
                continue
            if res == errno.ENODATA or nal >= end:
                break
            nal = min(nxt + 3, end)
        drain()
        L.edge264_free(C.byref(dec))
    finally:
        L.e264front_set_download(1)
    assert md5s(frames) == sums[name]["md5"]
