#!/usr/bin/env python3
"""tools/wire_probe.py -- where does a host-packet submission spend its time?  One 1080p fixture, N decoders, the front end's road (page-locked, trusted)
with version-4 and with wire packets: wall time per batch, host time inside the submit call, kernel time per launch (events), against resident packets.
usage (GPU box): python tools/wire_probe.py [file] [streams]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edge264_amd import backend, front, packet as P  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "nat1080_ipp30.264"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    data = open(os.path.join(ROOT, "tests", "golden", "streams", name), "rb").read()
    plain = [bytes(p) for p in front.capture_packets(data)[0]]
    wire = [bytes(p) for p in front.capture_packets(data, compact=True)[0]]
    front.capture_packets(b"", compact=False)
    dev = backend.Device(0)
    h0 = P.Packet(plain[0]).hdr
    W, H = int(h0["width_mbs"]), int(h0["height_mbs"])
    nb = int(h0["plane_size_Y"]) + int(h0["plane_size_C"])
    used = 0
    for p in plain:
        h = P.Packet(p).hdr
        used |= 1 << int(h["dst_slot"]) | int(h["ref_slots"])
    sts = []
    for _ in range(n):
        st = backend.Stream(dev, W, H)
        st.frame_bytes = nb
        for i in range(32):
            if used >> i & 1:
                st.alloc(i)
                st.fill(i, 0)
        sts.append(st)
    out = {"file": name, "streams": n}

    def timed(label, submit, count):
        for f in range(4 * count + 4):  # every slot of the batch ring has seen the largest batch (its staging buffer has grown) before the clock starts
            submit(f % count)
        dev.sync()
        dev.kernel_timing(True)
        host = 0.0
        t0 = time.perf_counter()
        for rep in range(2):
            for f in range(count):
                a = time.perf_counter()
                submit(f)
                host += time.perf_counter() - a
        dev.sync()
        wall = time.perf_counter() - t0
        k, l = dev.kernel_time_ms()
        dev.kernel_timing(False)
        out[label] = {"ms_per_batch_wall": round(1e3 * wall / (2 * count), 3), "ms_per_batch_inside_submit": round(1e3 * host / (2 * count), 3),
                      "kernel_ms_per_launch": [round(t / max(l, 1), 4) for t in k], "kernel_sum": round(sum(k) / max(l, 1), 3), "frames_per_s": round(2 * count * n / wall, 1)}

    dpk = [[dev.upload_packet(p) for p in plain] for _ in sts]
    bs = [dev.make_batch(sts, [dpk[k][f] for k in range(n)]) for f in range(len(plain))]
    timed("resident", lambda f: dev.submit_prepared(bs[f], backend.RUN_ALL), len(plain))
    # one page-locked buffer per stream and picture (what n decoders leave), both forms resident in host memory at once; the legs ALTERNATE so that a drift of the
    # box shows as a drift of both
    legs = {}
    for label, pk in (("pinned_v4", plain), ("pinned_wire", wire)):
        pins = [[dev.pinned_copy(p) for p in pk] for _ in range(n)]
        legs[label] = (pins, [dev.prepare_pinned_batch(sts, [pins[k][f] for k in range(n)], [len(pk[f])] * n) for f in range(len(pk))], pk)
    rounds = int(os.environ.get("E264_PROBE_ROUNDS", 3))
    for r in range(rounds):
        for label in ("pinned_v4", "pinned_wire"):
            pins, pbs, pk = legs[label]
            timed(f"{label}#{r}", lambda f: dev.submit_pinned_prepared(pbs[f], backend.RUN_ALL), len(pk))
            out[f"{label}#{r}"]["MB_per_batch"] = round(sum(len(p) for p in pk) * n / len(pk) / 1e6, 1)
    for label in legs:
        v = [out[f"{label}#{r}"]["frames_per_s"] for r in range(rounds)]
        out[label] = {"frames_per_s_rounds": v, "mean": round(sum(v) / len(v), 1), "min": min(v), "max": max(v)}
    for pins, _, _ in legs.values():
        for row in pins:
            for pp in row:
                dev.pinned_free(pp)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
