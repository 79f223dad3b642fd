#!/usr/bin/env python3
"""tools/pin_probe.py -- the host-packet entry points side by side on one 1080p fixture, 256 decoders, batch ring warmed: pageable packets (validated and gathered by the back end's host
threads), page-locked packets trusted / not trusted, one page-locked buffer per picture shared by the streams / one per stream.  usage (GPU box): python tools/pin_probe.py [file] [legs]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge264_amd import backend, front, packet as P  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "nat1080_ipp30.264"
    legs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["pageable", "pinned_trusted", "pinned_untrusted", "pinned_trusted_own_buffers", "pageable"]
    n = 256
    data = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "streams", name), "rb").read()
    plain = [bytes(p) for p in front.capture_packets(data)[0]]
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
    out = {"file": name, "host_threads": os.environ.get("E264_HOST_THREADS", "default")}

    def timed(label, submit, count):
        for f in range(4 * count + 4):
            submit(f % count)
        dev.sync()
        dev.kernel_timing(True)
        inside = 0.0
        t0 = time.perf_counter()
        for rep in range(3):
            for f in range(count):
                a = time.perf_counter()
                submit(f)
                inside += time.perf_counter() - a
        t_sub = time.perf_counter() - t0
        dev.sync()
        wall = time.perf_counter() - t0
        k, l = dev.kernel_time_ms()
        dev.kernel_timing(False)
        out.setdefault(label, []).append({"frames_per_s": round(3 * count * n / wall, 1), "ms_per_batch": round(1e3 * wall / (3 * count), 3), "ms_inside_submit": round(1e3 * inside / (3 * count), 3),
                                          "ms_until_last_submit_returns": round(1e3 * t_sub / (3 * count), 3), "kernel_ms": [round(t / max(l, 1), 4) for t in k]})
    hbs = pbs = pbs2 = None
    for leg in legs:
        if leg == "pageable":
            hbs = hbs or [dev.prepare_host_batch(sts, [plain[f]] * n) for f in range(len(plain))]
            timed(leg, lambda f: dev.submit_host_prepared(hbs[f], backend.RUN_ALL), len(plain))
        elif leg in ("pinned_trusted", "pinned_untrusted"):
            if pbs is None:
                pins = [dev.pinned_copy(p) for p in plain]
                pbs = [dev.prepare_pinned_batch(sts, [pins[f]] * n, [len(plain[f])] * n) for f in range(len(plain))]
            timed(leg, lambda f: dev.submit_pinned_prepared(pbs[f], backend.RUN_ALL, leg == "pinned_trusted"), len(plain))
        elif leg == "pinned_trusted_own_buffers":
            if pbs2 is None:
                pins2 = [[dev.pinned_copy(p) for p in plain] for _ in range(n)]
                pbs2 = [dev.prepare_pinned_batch(sts, [pins2[k][f] for k in range(n)], [len(plain[f])] * n) for f in range(len(plain))]
            timed(leg, lambda f: dev.submit_pinned_prepared(pbs2[f], backend.RUN_ALL, True), len(plain))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
