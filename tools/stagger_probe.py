#!/usr/bin/env python3
"""tools/stagger_probe.py -- what if the streams' GOPs are NOT in phase?  bench.py decodes picture f of every stream in submission f: all 256 I pictures share one
submission, in which the intra kernel (one workgroup per picture) keeps every CU busy for its 2.7 ms.  A fleet of independent streams has its I pictures anywhere:
here stream k runs `phase(k)` pictures ahead, so that every submission holds I and P pictures in the GOP's proportion.  Timing only (a stream that starts in the
middle of its GOP predicts from filled slots): resident packets, 256 streams, per-kernel times from events.
usage (GPU box): python tools/stagger_probe.py [streams] [lanes]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edge264_amd import backend, packet as P, synth  # noqa: E402

KERNELS = ("dbkparam2", "pred", "intra", "deblock")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    W, H, gop = 120, 68, "IPPPPPPP"
    V = 4
    vpk = [synth.StreamSynth(W, H, seed=1234 + v, t8x8=True, num_refs=2, residual_prob=0.3).gop(gop) for v in range(V)]
    G = len(gop)
    used = 0
    for pk in vpk:
        for p in pk:
            h = P.Packet(p).hdr
            used |= 1 << int(h["dst_slot"]) | int(h["ref_slots"])
    nb = P.frame_bytes(W, H)
    dev = backend.Device(0)
    dev.set_option("waves", 108)
    dev.set_option("intra_waves", 16)
    sts, dpk = [], []
    for k in range(n):
        st = backend.Stream(dev, W, H)
        st.frame_bytes = nb
        for i in range(max(used.bit_length(), 3)):
            st.alloc(i)
            st.fill(i, 128)
        st.bind_lane(k % lanes)
        sts.append(st)
        dpk.append([dev.upload_packet(p) for p in vpk[k % V]])
    groups = [list(range(g, n, lanes)) for g in range(lanes)]
    out = {"streams": n, "lanes": lanes, "gop": gop}
    for label, phase, split, planes in (("in_phase", lambda k: 0, 1, 1), ("staggered_lockstep", lambda k: (k // lanes) % G, 0, 0),
                                        ("staggered_one_workgroup", lambda k: (k // lanes) % G, 1, 0), ("staggered", lambda k: (k // lanes) % G, 1, 1),
                                        ("two_phases_lockstep", lambda k: ((k // lanes) % 2) * (G // 2), 0, 0), ("two_phases", lambda k: ((k // lanes) % 2) * (G // 2), 1, 1)):
        # split_intra 1 (the default): the I pictures' intra pass on the second queue from the start of a mixed submission; split_planes 1 (the default): their luma
        # and chroma on two workgroups (e264_intra_planes_kernel)
        dev.set_option("split_intra", split)
        dev.set_option("split_planes", planes)
        bs = [[dev.make_batch([sts[k] for k in idx], [dpk[k][(f + phase(k)) % G] for k in idx]) for idx in groups] for f in range(G)]
        for f in range(G):
            for b in bs[f]:
                dev.submit_prepared(b, backend.RUN_ALL)
        dev.sync()
        dev.kernel_timing(True)
        t0 = time.perf_counter()
        reps = 4
        for _ in range(reps):
            for f in range(G):
                for b in bs[f]:
                    dev.submit_prepared(b, backend.RUN_ALL)
        dev.sync()
        wall = time.perf_counter() - t0
        k4, l4 = dev.kernel_time_ms()
        dev.kernel_timing(False)
        out[label] = {"frames_per_s": round(reps * G * n / wall, 1), "ms_per_submission_wall": round(1e3 * wall / (reps * G * lanes), 3),
                      "kernel_ms_per_launch": {a: round(t / max(l4, 1), 4) for a, t in zip(KERNELS, k4)}}
        for row in bs:
            for b in row:
                dev.free_batch(b)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
