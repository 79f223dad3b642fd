#!/usr/bin/env python3
"""tools/check_gpu_corpus.py [seconds] [--wire] -- runs tools/_gpu_corpus.bin (tools/make_gpu_corpus.py) through the edge264.h API with the HIP sink on the device and compares
every NAL's return code and every frame's md5 with what the unmodified reference made of the same bytes in the build container."""
import hashlib
import json
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import HipFront  # noqa: E402 (test infrastructure: the ctypes binding of the edge264.h API; no oracle code runs here)

corpus = pickle.load(open(os.path.join(ROOT, "tools", "_gpu_corpus.bin"), "rb"))
h = HipFront()
h.lib.e264front_set_sink(0)
wire = "--wire" in sys.argv[1:]  # the front end folds its packets (include/edge264_compact.h), the device unfolds them
if wire:
    import ctypes
    h.lib.e264front_set_compact.argtypes = [ctypes.c_int]
    h.lib.e264front_set_compact(1)
t0 = time.time()
bad, pics, kinds = [], 0, {}
args = [a for a in sys.argv[1:] if a != "--wire"]
limit = float(args[0]) if args else 1e9  # seconds: stop cleanly and report what was checked
# interleave the kinds so that a time limit still samples all of them
order = sorted(range(len(corpus)), key=lambda i: (i % 13, i))
done = 0
for i in order:
    c = corpus[i]
    if time.time() - t0 > limit:
        break
    done += 1
    frames, codes = h.decode(c["data"])
    got = [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in frames]
    pics += len(got)
    k = kinds.setdefault(c["kind"], [0, 0])
    k[0] += 1
    if codes != c["codes"] or got != c["md5"]:
        k[1] += 1
        bad.append((c["kind"], c["seed"]))
print(json.dumps(dict(wire_form=wire, streams_in_corpus=len(corpus), streams=done, pictures=pics, mismatches=len(bad), first=bad[:10], per_kind={k: dict(streams=v[0], mismatches=v[1]) for k, v in kinds.items()},
                      seconds=round(time.time() - t0, 1))))
sys.exit(1 if bad else 0)
