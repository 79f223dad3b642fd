"""Capture replay: a capture file is a concatenation of command packets (self-describing through
E264FrameHdr.total_bytes), each tagged with the decoder it belongs to (E264FrameHdr.stream_id).  `e264_multi --dump-packets`
writes one; `Capture` reads it back and `replay()` feeds it to the MI355X back end in file order -- one batch per run of
packets of distinct streams -- so that kernels can be timed and re-verified on real motion / partition statistics without
the host entropy decoder in the loop (the reference's own trace output plays that role for its CPU path,
src/edge264_headers.c:1279-1281).

    python -m edge264_amd.replay capture.e264            # per-picture md5 of every stream, decode order
    python bench.py --capture capture.e264               # the capture's first stream as the benchmark GOP
"""
from __future__ import annotations

import hashlib
import sys

import numpy as np

from . import packet as P


class Capture:
    def __init__(self, data: bytes):
        self.packets: list[bytes] = []
        off = 0
        while off < len(data):
            if len(data) - off < P.FRAME_HDR.itemsize:
                raise ValueError("truncated capture file")
            hdr = np.frombuffer(data, P.FRAME_HDR, 1, off)[0]
            n = int(hdr["total_bytes"])
            if int(hdr["magic"]) != P.E264_MAGIC or n < P.FRAME_HDR.itemsize or off + n > len(data):
                raise ValueError(f"not a command packet at offset {off}")
            self.packets.append(bytes(data[off:off + n]))
            off += n

    @classmethod
    def load(cls, path: str) -> "Capture":
        with open(path, "rb") as f:
            return cls(f.read())

    @staticmethod
    def write(path: str, packets, stream_ids=None) -> None:
        with open(path, "wb") as f:
            for i, p in enumerate(packets):
                b = bytearray(p)
                if stream_ids is not None:
                    np.frombuffer(b, P.FRAME_HDR, 1)["stream_id"] = stream_ids[i]
                f.write(b)

    def stream_ids(self) -> list[int]:
        return [int(np.frombuffer(p, P.FRAME_HDR, 1)[0]["stream_id"]) for p in self.packets]

    def of_stream(self, sid: int) -> list[bytes]:
        return [p for p, s in zip(self.packets, self.stream_ids()) if s == sid]


def batches(cap: Capture):
    """File order cut into runs of packets of pairwise distinct streams (= what one submission may hold)."""
    run, seen = [], set()
    for p, sid in zip(cap.packets, cap.stream_ids()):
        if sid in seen:
            yield run
            run, seen = [], set()
        run.append((sid, p))
        seen.add(sid)
    if run:
        yield run


def replay(cap: Capture, device, on_picture=None, mode: int = 3) -> dict[int, list[str]]:
    """Runs the capture on `device` (edge264_amd.backend.Device).  Returns {stream_id: [md5 of every decoded picture, decode
    order]} (the whole slot: planar Y then CbCr rows, the reference's frame layout); on_picture(sid, packet, samples) sees
    every picture too."""
    from . import backend
    streams: dict[int, backend.Stream] = {}
    allocated: dict[int, set] = {}
    out: dict[int, list[str]] = {}
    try:
        for run in batches(cap):
            sts, dps = [], []
            for sid, pkt in run:
                h = P.Packet(pkt).hdr
                if sid not in streams:
                    streams[sid] = backend.Stream(device, int(h["width_mbs"]), int(h["height_mbs"]))
                    allocated[sid] = set()
                st = streams[sid]
                st.frame_bytes = int(h["plane_size_Y"]) + int(h["plane_size_C"])
                for s in range(P.MAX_SLOTS):
                    if (s == int(h["dst_slot"]) or int(h["ref_slots"]) >> s & 1) and s not in allocated[sid]:
                        st.alloc(s)
                        st.fill(s, 0)  # like the front end's frame_fill(0) for pictures that are referenced before they are decoded
                        allocated[sid].add(s)
                sts.append(st)
                dps.append(device.upload_packet(pkt))
            device.submit_batch(sts, dps, mode)
            for (sid, pkt), st, dp in zip(run, sts, dps):
                samples = st.download(int(P.Packet(pkt).hdr["dst_slot"]))
                out.setdefault(sid, []).append(hashlib.md5(samples.tobytes()).hexdigest())
                if on_picture:
                    on_picture(sid, pkt, samples)
                dp.free()
    finally:
        for st in streams.values():
            st.close()
    return out


def main(argv=None) -> int:
    import json
    from . import backend
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        print(__doc__)
        return 2
    dev = backend.Device(0)  # no CPU fallback: raises without the HIP library / a GPU
    try:
        res = replay(Capture.load(argv[0]), dev)
    finally:
        dev.close()
    print(json.dumps({str(k): v for k, v in sorted(res.items())}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
