"""GPU parity: the HIP back end, called through the C ABI (libedge264_hip.so), against the
CPU oracle on the same seeded command packets.  Bit-exact or fail (integer/byte work)."""
import numpy as np
import pytest

from edge264_amd import packet as P
from edge264_amd import synth

pytestmark = pytest.mark.gpu

ALL_I = (P.MB_I8x8, P.MB_I4x4, P.MB_I16x16)
CASES = [
    ("intra4x4_16x16", "III", dict()),
    ("intra8x8", "III", dict(i_kinds=ALL_I, t8x8=True)),
    ("pcm", "II", dict(pcm_prob=0.2)),
    ("scaling_lists", "II", dict(scaling=True, i_kinds=ALL_I)),
    ("ippp", "IPPPP", dict()),
    ("ippp_t8x8_scaling", "IPPPP", dict(t8x8=True, scaling=True)),
    ("ippp_explicit_wp", "IPPPP", dict(weighted=1)),
    ("ibbp", "IPBBPBB", dict()),
    ("ibbp_explicit_wp", "IPBBPBB", dict(weighted=1, t8x8=True)),
    ("ibbp_implicit_wp", "IPBBPBB", dict(weighted=2, t8x8=True, scaling=True)),
    ("slices_idc2", "IPBBP", dict(slices_per_frame=4, deblock_idc=2)),
    ("slices_idc0", "IPBBP", dict(slices_per_frame=4, deblock_idc=0)),
    # every slice with its own scaling lists and weight tables: the per-wave LDS slice cache is reloaded inside a strip
    ("slices_scaling_explicit", "IPBBP", dict(slices_per_frame=6, weighted=1, scaling=True, t8x8=True, i_kinds=ALL_I)),
    ("slices_scaling_implicit", "IPBBP", dict(slices_per_frame=6, weighted=2, scaling=True, t8x8=True, i_kinds=ALL_I)),
    ("filter_offsets", "IPB", dict(filter_offsets=(6, -4))),
    ("no_deblock", "IPB", dict(deblock=False)),
    ("stress_explicit", "IPBBP", dict(stress=True, weighted=1, t8x8=True, scaling=True, i_kinds=ALL_I)),
    ("stress_implicit_far_mv", "IPBBP", dict(stress=True, weighted=2, t8x8=True, mv_range=400)),
]


@pytest.fixture(scope="module")
def device():
    from edge264_amd import backend
    dev = backend.Device(0)  # raises if the extension or the GPU is missing: no silent fallback
    yield dev
    dev.close()


def describe_mismatch(pk, a, b, w, h):
    g = P.frame_geometry(w, h)
    diff = np.nonzero(a != b)[0]
    x = int(diff[0])
    if x < g["plane_size_Y"]:
        yy, xx = divmod(x, g["stride_Y"])
        mb = pk.mbs[(yy // 16) * w + xx // 16]
        return f"{len(diff)} bytes differ; first luma ({xx},{yy}) mb({xx // 16},{yy // 16}) kind {mb['kind']} flags {mb['flags']} hip {a[x]} oracle {b[x]}"
    yy, xx = divmod(x - g["plane_size_Y"], g["stride_C"])
    half = g["stride_C"] // 2
    mb = pk.mbs[(yy // 8) * w + (xx % half) // 8]
    return f"{len(diff)} bytes differ; first chroma plane {xx // half} ({xx % half},{yy}) kind {mb['kind']} hip {a[x]} oracle {b[x]}"


def run_stream(device, oracle, seed, pattern, kw, w, h, passes_split=True):
    from edge264_amd import backend
    s = synth.StreamSynth(w, h, seed, **kw)
    nb = P.frame_bytes(w, h)
    rng = np.random.default_rng(seed + 1000)
    ns = kw.get("n_slots", 6)
    dpb = [rng.integers(0, 256, nb + 16, dtype=np.uint8) for _ in range(ns)] + [None] * (32 - ns)
    st = backend.Stream(device, w, h)
    try:
        for i in range(ns):
            st.alloc(i)
            st.upload(i, dpb[i][:nb])
        for i, t in enumerate(pattern):
            pkt = s.next_frame(t)
            pk = P.Packet(pkt)
            d = int(pk.hdr["dst_slot"])
            if passes_split:
                dp = device.upload_packet(pkt)
                for passes in (1, 2):
                    oracle.decode_frame(pkt, dpb, passes)
                    device.submit_batch([st], [dp], passes)
                    got = st.download(d)
                    assert np.array_equal(got, dpb[d][:nb]), f"seed {seed} frame {i}{t} pass {passes}: " + describe_mismatch(pk, got, dpb[d][:nb], w, h)
                dp.free()
            else:
                oracle.decode_frame(pkt, dpb, 3)
                st.submit(pkt)  # the front-end path: host packet -> pinned staging -> H2D -> kernels
                got = st.download(d)
                assert np.array_equal(got, dpb[d][:nb]), f"seed {seed} frame {i}{t}: " + describe_mismatch(pk, got, dpb[d][:nb], w, h)
    finally:
        st.close()


@pytest.mark.parametrize("name,pattern,kw", CASES, ids=[c[0] for c in CASES])
def test_hip_matches_oracle(device, oracle, name, pattern, kw):
    for seed in range(2):
        run_stream(device, oracle, seed, pattern, kw, 6, 5)


def test_submit_path(device, oracle):
    run_stream(device, oracle, 5, "IPBBP", dict(t8x8=True, weighted=1, i_kinds=ALL_I), 7, 4, passes_split=False)


def test_odd_geometry(device, oracle):
    for (w, h) in ((1, 1), (1, 4), (5, 1), (2, 2), (3, 19)):
        run_stream(device, oracle, 7, "IPB", dict(t8x8=True, i_kinds=ALL_I, mv_range=100), w, h)


def test_submit_batch_host(device, oracle):
    """e264hip_submit_batch_host: host packets of several streams, staged + copied + launched asynchronously,
    several rounds in flight before anything is read back."""
    from edge264_amd import backend
    w, h, n_streams = 9, 5, 5
    gens = [synth.StreamSynth(w, h, 50 + k, t8x8=bool(k & 1), i_kinds=ALL_I) for k in range(n_streams)]
    nb = P.frame_bytes(w, h)
    dpbs = [[np.full(nb + 16, 128, np.uint8) for _ in range(6)] + [None] * 26 for _ in range(n_streams)]
    sts = [backend.Stream(device, w, h) for _ in range(n_streams)]
    try:
        for st in sts:
            for i in range(6):
                st.alloc(i)
                st.fill(i, 128)
        last = [0] * n_streams
        for t in "IPBPB":   # five rounds, no synchronisation in between
            pkts = [g.next_frame(t) for g in gens]
            for k, pkt in enumerate(pkts):
                oracle.decode_frame(pkt, dpbs[k], 3)
                last[k] = int(P.Packet(pkt).hdr["dst_slot"])
            device.submit_batch_host(sts, pkts)
        for k, st in enumerate(sts):
            for slot in range(6):
                assert np.array_equal(st.download(slot), dpbs[k][slot][:nb]), f"stream {k} slot {slot}"
    finally:
        for st in sts:
            st.close()


def test_padded_strides(device, oracle):
    """Frames whose strides carry the reference's anti-aliasing padding (src/edge264_headers.c:2032-2041):
    2048 luma samples wide -> stride_Y + 16; 4096 wide -> stride_C + 8, i.e. Cr rows only 4-byte aligned
    (the 16-byte vector stores of the strip / group flushes must cope)."""
    run_stream(device, oracle, 11, "IPB", dict(i_kinds=ALL_I), 128, 2)
    run_stream(device, oracle, 12, "IPP", dict(t8x8=True), 256, 2)


def test_1080p_ipb(device, oracle):
    """BASELINE geometry (120 x 68 macroblocks) with B frames, 8x8 transform and weighted prediction."""
    run_stream(device, oracle, 21, "IPB", dict(t8x8=True, weighted=1, i_kinds=ALL_I), 120, 68, passes_split=False)


def test_1080p_config3(device, oracle):
    """BASELINE configs[3] at its own size: IBBP, 8x8 transform, scaling lists, explicit weighted bi-prediction, two
    references, deblocking -- four seeds, every frame compared."""
    for seed in (31, 32, 33, 34):
        run_stream(device, oracle, seed, "IPBBP", dict(t8x8=True, scaling=True, weighted=1, num_refs=2, i_kinds=ALL_I), 120, 68, passes_split=False)


def test_qp_range(device, oracle):
    """Every qP % 6 / qP / 6 combination of the dequantisers (normAdjust is arithmetic on immediates in the kernels), both
    transforms, custom scaling lists, and the alpha / beta / tC0 table ends of the deblocking filter."""
    for qp in range(0, 52, 3):
        run_stream(device, oracle, 60 + qp, "IPB", dict(qp_base=qp, t8x8=True, scaling=True, i_kinds=ALL_I, residual_prob=0.8), 4, 3, passes_split=False)


def test_many_references(device, oracle):
    """16 reference pictures per list, 17 DPB slots in use: reference indices up to 15 into the explicit / implicit weight
    tables ([LX * 32 + refIdx], [refIdxL0][refIdxL1]) and the slot table."""
    run_stream(device, oracle, 51, "IPPPPPPPPPPPPPPPPBPB", dict(num_refs=16, n_slots=17, weighted=1, i_kinds=ALL_I, t8x8=True), 5, 4, passes_split=False)
    run_stream(device, oracle, 52, "IPPPPPPPPPPPPPPPPBPB", dict(num_refs=16, n_slots=17, weighted=2), 5, 4, passes_split=False)


def test_missing_macroblocks(device, oracle):
    """Packets of incomplete pictures (a lost slice leaves its macroblocks E264_MB_ABSENT, include/edge264_cmd.h): absent
    macroblocks are not written by any kernel and not deblocked, their neighbours still are; a picture with nothing
    in it leaves the slot untouched."""
    w, h = 9, 6
    nb = P.frame_bytes(w, h)
    for seed, pattern in ((41, "IPB"), (42, "IPP")):
        s = synth.StreamSynth(w, h, seed, t8x8=True, i_kinds=ALL_I)
        rng = np.random.default_rng(seed)
        dpb = [rng.integers(0, 256, nb + 16, dtype=np.uint8) for _ in range(6)] + [None] * 26
        st = __import__("edge264_amd.backend", fromlist=["Stream"]).Stream(device, w, h)
        try:
            for i in range(6):
                st.alloc(i)
                st.upload(i, dpb[i][:nb])
            for i, t in enumerate(pattern):
                buf = bytearray(s.next_frame(t))
                pk = P.Packet(buf)
                mbs = np.frombuffer(buf, P.MB, w * h, int(pk.hdr["mbs_off"]))
                if i == len(pattern) - 1:
                    mbs["kind"][:] = P.MB_ABSENT          # nothing arrived
                else:
                    a = int(rng.integers(0, w * h - 8))
                    mbs["kind"][a:a + int(rng.integers(3, 2 * w))] = P.MB_ABSENT  # a lost slice
                P.refresh_summary(buf)  # the emitter counts what it emitted
                pkt = bytes(buf)
                d = int(pk.hdr["dst_slot"])
                before = dpb[d].copy()
                oracle.decode_frame(pkt, dpb, 3)
                st.submit(pkt)
                got = st.download(d)
                assert np.array_equal(got, dpb[d][:nb]), f"seed {seed} frame {i}{t}: " + describe_mismatch(pk, got, dpb[d][:nb], w, h)
                if i == len(pattern) - 1:
                    assert np.array_equal(got, before[:nb])
        finally:
            st.close()


def test_max_frame_size(device, oracle):
    """Level 5.1/5.2 maximum picture (4096 x 2304 = 256 x 144 macroblocks, 36 864 MBs): four 64-macroblock scan
    chunks per row in the intra kernel, 6 rounds of row pairs in the deblocking kernel, 32-bit offsets into a 14 MB plane."""
    run_stream(device, oracle, 31, "IPB", dict(t8x8=True, i_kinds=ALL_I), 256, 144, passes_split=False)


@pytest.mark.parametrize("waves", [2, 4, 7, 8, 106, 108])
def test_deblock_waves_per_frame(device, oracle, waves):
    """Frames wider than the LDS strips and taller than one round of the deblocking kernel (5 x waves rows): strip wrap,
    the hand-off between waves and between rounds through memory, a last group of fewer than five rows.  100 + n: the kernel with
    luma waves (8 rows) and chroma waves (16 rows), n waves per picture."""
    prev = device.set_option("waves", waves)
    try:
        run_stream(device, oracle, 3, "IPB", dict(t8x8=True, i_kinds=ALL_I), 5, 21)
        run_stream(device, oracle, 4, "IPP", dict(), 26, 5 * (waves % 100) * 2 + 3)
    finally:
        device.set_option("waves", prev)


@pytest.mark.parametrize("waves", [4, 8, 16])
def test_intra_waves_per_frame(device, oracle, waves):
    """Frames taller than one round of the intra wavefront kernel (one macroblock row per wave)."""
    prev = device.set_option("intra_waves", waves)
    try:
        run_stream(device, oracle, 5, "IPB", dict(t8x8=True, i_kinds=ALL_I), 5, 21)
        run_stream(device, oracle, 6, "IIP", dict(i_kinds=ALL_I), 26, 2 * waves + 3)
    finally:
        device.set_option("intra_waves", prev)


def test_geometry_and_batch_validation(device):
    """A packet may not name a picture larger than the slots it writes / reads, a batch may not hold a stream twice nor
    refer to unallocated reference slots: EINVAL from the C ABI, never a GPU fault (round-1 advisor findings)."""
    from edge264_amd import backend
    w, h = 4, 3
    g = synth.StreamSynth(w, h, 3)
    i_pkt, p_pkt = g.next_frame("I"), g.next_frame("P")
    nb = P.frame_bytes(w, h)
    st, st2 = backend.Stream(device, w, h), backend.Stream(device, w, h)
    try:
        small = nb // 2
        for s_ in (st, st2):
            for i in range(6):
                s_.L.e264hip_frame_alloc(s_.h, i, small if s_ is st else nb, None)
        # 1. destination slot smaller than the picture: front-end path and batch path
        with pytest.raises(backend.BackendError):
            st.submit(i_pkt)
        di, dp = device.upload_packet(i_pkt), device.upload_packet(p_pkt)
        with pytest.raises(backend.BackendError):
            device.submit_batch([st], [di], backend.RUN_ALL)
        # 2. the same stream twice in one batch
        with pytest.raises(backend.BackendError):
            device.submit_batch([st2, st2], [di, di], backend.RUN_ALL)
        # 3. reference slot of the P frame not allocated in this stream
        ref = int(np.nonzero(P.Packet(p_pkt).motion["refPic"].max(axis=0) >= 0)[0][0]) if False else None
        used = sorted({int(x) for x in P.Packet(p_pkt).motion["refPic"].ravel() if x >= 0})
        assert used
        st2.free(used[0])
        with pytest.raises(backend.BackendError):
            device.submit_batch([st2], [dp], backend.RUN_ALL)
        # the good path still works afterwards
        st2.alloc(used[0])
        st2.fill(used[0], 128)
        device.submit_batch([st2], [di], backend.RUN_ALL)
        di.free(); dp.free()
    finally:
        st.close(); st2.close()


def test_host_batch_is_vetted_on_every_thread(device, oracle):
    """A batch from pageable memory is validated and staged by the back end's pool of host threads: a corrupt packet
    anywhere in it is EINVAL with the worker's message, nothing of the batch is submitted, and a good batch of the same
    size decodes bit-exactly afterwards."""
    from edge264_amd import backend
    w, h, n_streams = 7, 4, 12   # >= 4 packets: the pool is used
    gens = [synth.StreamSynth(w, h, 80 + k, i_kinds=ALL_I) for k in range(n_streams)]
    nb = P.frame_bytes(w, h)
    dpbs = [[np.full(nb + 16, 128, np.uint8) for _ in range(6)] + [None] * 26 for _ in range(n_streams)]
    sts = [backend.Stream(device, w, h) for _ in range(n_streams)]
    try:
        for st in sts:
            for i in range(6):
                st.alloc(i)
                st.fill(i, 128)
        pkts = [g.next_frame("I") for g in gens]
        for bad_at in (0, 5, n_streams - 1):
            broken = list(pkts)
            raw = bytearray(pkts[bad_at])
            mbs_off = int(P.Packet(pkts[bad_at]).hdr["mbs_off"])
            raw[mbs_off + 3 * 32] = 9                      # macroblock kind out of range
            broken[bad_at] = bytes(raw)
            with pytest.raises(backend.BackendError, match="macroblock kind"):
                device.submit_batch_host(sts, broken)
        device.sync()
        for st in sts:                                     # nothing was written
            assert (st.download(int(P.Packet(pkts[0]).hdr["dst_slot"])) == 128).all()
        for k, pkt in enumerate(pkts):
            oracle.decode_frame(pkt, dpbs[k], 3)
        device.submit_batch_host(sts, pkts)
        for k, st in enumerate(sts):
            slot = int(P.Packet(pkts[k]).hdr["dst_slot"])
            assert np.array_equal(st.download(slot), dpbs[k][slot][:nb]), f"stream {k}"
    finally:
        for st in sts:
            st.close()
