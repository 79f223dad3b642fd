#!/usr/bin/env python3
"""Debug helper: one stream, frame by frame, HIP vs oracle at an arbitrary size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from edge264_amd import backend, packet as P, synth
from oracle.pyoracle import Oracle

w, h, pattern = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
kw = eval(sys.argv[4]) if len(sys.argv) > 4 else {}
dev = backend.Device(0)
st = backend.Stream(dev, w, h)
orc = Oracle()
nb = P.frame_bytes(w, h)
dpb = [np.full(nb + 16, 128, np.uint8) for _ in range(6)] + [None] * 26
for i in range(6):
    st.alloc(i); st.upload(i, dpb[i][:nb])
s = synth.StreamSynth(w, h, 1234, **kw)
for i, t in enumerate(pattern):
    pkt = s.next_frame(t)
    pk = P.Packet(pkt)
    d = int(pk.hdr["dst_slot"])
    dp = dev.upload_packet(pkt)
    for passes in (1, 2):
        orc.decode_frame(pkt, dpb, passes)
        t0 = time.time()
        dev.submit_batch([st], [dp], passes)
        got = st.download(d)
        ok = np.array_equal(got, dpb[d][:nb])
        print(f"frame {i}{t} pass {passes}: {'OK' if ok else 'MISMATCH ' + str(int((got != dpb[d][:nb]).sum()))} ({(time.time()-t0)*1e3:.1f} ms)", flush=True)
    dp.free()
