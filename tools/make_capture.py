#!/usr/bin/env python3
"""tools/make_capture.py a.264 [b.264 ...] out.e264 -- decodes Annex-B streams with the reference's front end bound to our
packet emitters (capture sink, CPU only: edge264_amd/libedge264_hipfront.so) and writes the command packets as a capture
file, streams interleaved round-robin and tagged with their index.  `python -m edge264_amd.replay out.e264` replays it on
the GPU, `python bench.py --capture out.e264` times it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge264_amd import replay  # noqa: E402
from oracle.pyoracle import HipFront, Oracle  # noqa: E402  (the capture sink hands packets back through the checker-side binding)


def main():
    *ins, out = sys.argv[1:]
    orc = Oracle()
    per = [HipFront().decode_capture(open(p, "rb").read(), orc)[2] for p in ins]
    packets, sids = [], []
    for i in range(max(len(p) for p in per)):
        for sid, pk in enumerate(per):
            if i < len(pk):
                packets.append(pk[i])
                sids.append(sid)
    replay.Capture.write(out, packets, sids)
    print(f"{out}: {len(packets)} packets of {len(per)} streams, {os.path.getsize(out)} bytes")


if __name__ == "__main__":
    main()
