"""Host side of the back end on the MI355X (round 3): compute lanes, the upload queue, the memory recycler and the
trusted submission path -- everything bit-exact against the oracle, nothing waits for or disturbs another decoder."""
import ctypes as C
import errno

import numpy as np
import pytest

from edge264_amd import backend, packet as P, synth

pytestmark = pytest.mark.gpu
ALL_I = (P.MB_I4x4, P.MB_I8x8, P.MB_I16x16)


@pytest.fixture(scope="module")
def device():
    dev = backend.Device(0)
    yield dev
    dev.close()


def _streams(device, n, w, h, slots=6):
    sts = [backend.Stream(device, w, h) for _ in range(n)]
    for st in sts:
        for i in range(slots):
            st.alloc(i)
            st.fill(i, 128)
    return sts


def test_lanes_overlap_and_stay_bit_exact(device, oracle):
    """Streams bound to four different compute lanes, ten rounds queued on every lane without any synchronisation in between
    (resident batches and host batches mixed): every slot of every stream equals the oracle's; a batch that mixes lanes and a
    lane out of range are EINVAL."""
    w, h, n = 11, 7, 8
    nb = P.frame_bytes(w, h)
    gens = [synth.StreamSynth(w, h, 300 + k, t8x8=bool(k & 1), i_kinds=ALL_I, num_refs=2) for k in range(n)]
    dpbs = [[np.full(nb + 16, 128, np.uint8) for _ in range(6)] + [None] * 26 for _ in range(n)]
    sts = _streams(device, n, w, h)
    try:
        for k, st in enumerate(sts):
            st.bind_lane(k % backend.MAX_LANES)
        with pytest.raises(backend.BackendError):
            sts[0].bind_lane(backend.MAX_LANES)
        groups = [[k for k in range(n) if k % backend.MAX_LANES == g] for g in range(backend.MAX_LANES)]
        keep = []
        for r, t in enumerate("IPBPBPPBPP"):
            pkts = [g.next_frame(t) for g in gens]
            for k, pkt in enumerate(pkts):
                oracle.decode_frame(pkt, dpbs[k], 3)
            for idx in groups:
                if r & 1:   # host packets: staged, copied on the upload queue, launched on the lane
                    device.submit_batch_host([sts[k] for k in idx], [pkts[k] for k in idx])
                else:       # packets resident in HBM
                    dp = [device.upload_packet(pkts[k]) for k in idx]
                    b = device.make_batch([sts[k] for k in idx], dp)
                    device.submit_prepared(b)
                    keep.append((b, dp))
            if r == 4:
                with pytest.raises(backend.BackendError, match="one compute lane"):
                    device.submit_batch_host([sts[0], sts[1]], [pkts[0], pkts[1]])
        for k, st in enumerate(sts):
            for slot in range(6):
                assert np.array_equal(st.download(slot), dpbs[k][slot][:nb]), f"stream {k} slot {slot}"
        for b, dp in keep:
            device.free_batch(b)
            for d in dp:
                d.free()
    finally:
        for st in sts:
            st.close()


def test_one_decoder_reallocates_while_the_others_run(device, oracle):
    """63 streams keep submitting; one stream frees and reallocates its slots (an SPS change), closes and is replaced by a
    new one between their rounds -- frame_alloc / frame_free / stream_close / stream_open never drain the device (the memory
    is parked and recycled), and nothing the others decode is disturbed."""
    w, h, n = 6, 4, 64
    nb = P.frame_bytes(w, h)
    gens = [synth.StreamSynth(w, h, 500 + k, i_kinds=ALL_I) for k in range(n)]
    dpbs = [[np.full(nb + 16, 128, np.uint8) for _ in range(6)] + [None] * 26 for _ in range(n)]
    sts = _streams(device, n, w, h)
    try:
        for r, t in enumerate("IPPBPPBP"):
            pkts = [g.next_frame(t) for g in gens]
            for k in range(1, n):
                oracle.decode_frame(pkts[k], dpbs[k], 3)
            device.submit_batch_host(sts[1:], pkts[1:])
            # stream 0: new picture size -> every slot given back and allocated again, twice; then the decoder itself goes
            odd = sts[0]
            for _ in range(2):
                for i in range(6):
                    odd.free(i)
                for i in range(6):
                    odd.alloc(i)
                    odd.fill(i, 128)
            if r in (2, 5):
                odd.close()
                sts[0] = backend.Stream(device, w, h)
                for i in range(6):
                    sts[0].alloc(i)
                    sts[0].fill(i, 128)
                gens[0] = synth.StreamSynth(w, h, 900 + r, i_kinds=ALL_I)
                dpbs[0] = [np.full(nb + 16, 128, np.uint8) for _ in range(6)] + [None] * 26
            else:
                for d in dpbs[0][:6]:
                    d[:] = 128
                gens[0] = synth.StreamSynth(w, h, 900 + r, i_kinds=ALL_I)
            p0 = gens[0].next_frame("I")
            oracle.decode_frame(p0, dpbs[0], 3)
            sts[0].submit(p0)
            d0 = int(P.Packet(p0).hdr["dst_slot"])
            assert np.array_equal(sts[0].download(d0), dpbs[0][d0][:nb]), f"round {r}: the reallocating stream"
        for k in range(1, n):
            for slot in range(6):
                assert np.array_equal(sts[k].download(slot), dpbs[k][slot][:nb]), f"stream {k} slot {slot}"
    finally:
        for st in sts:
            st.close()


def test_trusted_packets_are_still_held_against_the_streams_slots(device, oracle):
    """E264_SUBMIT_TRUSTED skips the per-macroblock walk, not the slots: a vetted packet whose (verified) header names a
    reference slot this stream has not allocated, or a picture larger than a slot, is EINVAL -- never a GPU fault (round-2
    advisor finding); a header whose ref_slots disagrees with the motion records does not pass packet_check in the first place."""
    w, h = 6, 4
    nb = P.frame_bytes(w, h)
    g = synth.StreamSynth(w, h, 77, num_refs=2)
    i_pkt, p_pkt = g.next_frame("I"), g.next_frame("P")
    assert backend.packet_check(p_pkt) == 0
    lying = bytearray(p_pkt)
    np.frombuffer(lying, P.FRAME_HDR, 1)["ref_slots"] = 0
    assert backend.packet_check(bytes(lying)) == errno.EINVAL
    st = backend.Stream(device, w, h)
    try:
        for i in range(6):
            st.alloc(i)
            st.fill(i, 128)
        pins = [device.pinned_copy(p) for p in (i_pkt, p_pkt)]
        dpb = [np.full(nb + 16, 128, np.uint8) for _ in range(6)] + [None] * 26
        device.submit_pinned_prepared(device.prepare_pinned_batch([st], [pins[0]], [len(i_pkt)]))
        oracle.decode_frame(i_pkt, dpb, 3)
        d = int(P.Packet(i_pkt).hdr["dst_slot"])
        assert np.array_equal(st.download(d), dpb[d][:nb])  # the trusted path decodes
        used = [s for s in range(32) if int(P.Packet(p_pkt).hdr["ref_slots"]) >> s & 1]
        st.free(used[0])                                   # the slot the vetted packet predicts from is gone
        with pytest.raises(backend.BackendError, match="reference slot not allocated"):
            device.submit_pinned_prepared(device.prepare_pinned_batch([st], [pins[1]], [len(p_pkt)]))
        st.L.e264hip_frame_alloc(st.h, used[0], nb // 2, None)   # there again, but too small for the picture
        with pytest.raises(backend.BackendError, match="larger than a reference slot"):
            device.submit_pinned_prepared(device.prepare_pinned_batch([st], [pins[1]], [len(p_pkt)]))
        device.sync()
        st.free(used[0])
        st.alloc(used[0])                                  # the right size again: the same vetted packet now goes through
        st.upload(used[0], dpb[used[0]][:nb])
        device.submit_pinned_prepared(device.prepare_pinned_batch([st], [pins[1]], [len(p_pkt)]))
        oracle.decode_frame(p_pkt, dpb, 3)
        d = int(P.Packet(p_pkt).hdr["dst_slot"])
        assert np.array_equal(st.download(d), dpb[d][:nb])
        for p in pins:
            device.pinned_free(p)
    finally:
        st.close()


def test_upload_queue_orders_copies_and_kernels(device, oracle):
    """Twelve host batches in a row on the upload queue (more than the staging rings are deep, so slots are reused while
    earlier batches are still in flight), with the option off for comparison: the same, bit-exact results."""
    w, h, n = 20, 12, 6
    nb = P.frame_bytes(w, h)
    for up in (1, 0):
        prev = device.set_option("upload_queue", up)
        gens = [synth.StreamSynth(w, h, 700 + k, t8x8=True, i_kinds=ALL_I) for k in range(n)]
        dpbs = [[np.full(nb + 16, 128, np.uint8) for _ in range(6)] + [None] * 26 for _ in range(n)]
        sts = _streams(device, n, w, h)
        try:
            for t in "IPPBPPBPPBPP":
                pkts = [g.next_frame(t) for g in gens]
                for k, pkt in enumerate(pkts):
                    oracle.decode_frame(pkt, dpbs[k], 3)
                device.submit_batch_host(sts, pkts)
            for k, st in enumerate(sts):
                for slot in range(6):
                    assert np.array_equal(st.download(slot), dpbs[k][slot][:nb]), f"upload_queue={up} stream {k} slot {slot}"
        finally:
            for st in sts:
                st.close()
            device.set_option("upload_queue", prev)


def test_event_query_never_blocks_and_settles(device, oracle):
    """e264hip_event_query (how a front end that submits packets in place learns that their buffers may be reused): EBUSY or 0
    right after the record, 0 once the device has been waited for; an index outside the 16 slots is refused."""
    w, h, n = 20, 12, 4
    nb = P.frame_bytes(w, h)
    gens = [synth.StreamSynth(w, h, 900 + k, i_kinds=ALL_I) for k in range(n)]
    dpbs = [[np.full(nb + 16, 128, np.uint8) for _ in range(6)] + [None] * 26 for _ in range(n)]
    sts = _streams(device, n, w, h)
    try:
        for t in "IPP":
            pkts = [g.next_frame(t) for g in gens]
            for k, pkt in enumerate(pkts):
                oracle.decode_frame(pkt, dpbs[k], 3)
            device.submit_batch_host(sts, pkts)
        device.event_record(9)
        first = device.L.e264hip_event_query(device.h, 9)
        assert first in (0, errno.EBUSY)
        device.sync()
        assert device.event_done(9)
        assert device.L.e264hip_event_query(device.h, 16) == errno.EINVAL
        assert device.L.e264hip_event_query(device.h, -1) == errno.EINVAL
        for k, st in enumerate(sts):
            for slot in range(6):
                assert np.array_equal(st.download(slot), dpbs[k][slot][:nb]), f"stream {k} slot {slot}"
    finally:
        for st in sts:
            st.close()


@pytest.mark.gpu
def test_bench_distributed_path_on_one_gpu():
    """The multi-GPU code path of bench.py on the one GPU there is: E264_FORCE_DIST=1 makes
    rank 0 of a world of 1 create the RCCL process group (backend "nccl", device_id), run the barrier and the all_gather / all_reduce
    of the rates, and bind itself to the CPUs of the GPU's NUMA node like every rank of an 8-GPU run does."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, E264_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--streams", "4", "--no-cpu-baseline",
                        "--no-other-configs", "--no-host-packets", "--no-same-input"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["bit_exact"] is True
    assert line["per_rank"]["frames_per_s"] and line["per_rank"]["min"] > 0
    numa = line["per_rank"]["numa"]
    assert set(numa) >= {"numa_node", "bound", "cpus"}
    if os.path.isdir("/sys/devices/system/node/node0") and numa["numa_node"] < 0:
        # the platform has NUMA nodes in sysfs: the GPU's node must have been found through its PCI address
        pytest.fail(f"NUMA node of the GPU not found: {numa}")
