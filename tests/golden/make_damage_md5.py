#!/usr/bin/env python3
"""Writes tests/golden/streams/damage_md5.json: what the UNMODIFIED reference decoder (oracle/_ref/libedge264_ref.so, built
from /root/reference) outputs for the damaged-stream scenarios of tests/damage.py.  The GPU box has no reference: the GPU
test compares the HIP sink with these md5s.  Run in the container: python tests/golden/make_damage_md5.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import ref_decoder  # noqa: E402
from tests import damage  # noqa: E402

ref = ref_decoder()
out = {}
for name, which, keep in damage.RESENT:
    frames, codes = ref.decode(damage.truncated_then_resent(name, which, keep))
    out[f"{name}-{which}-{keep}"] = {"nal_codes": codes, "md5": [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in frames]}
for name, which, ka, kb in damage.RESENT2:
    frames, codes = ref.decode(damage.two_truncated_then_resent(name, which, ka, kb))
    out[f"{name}-{which}+{which + 1}-{ka}-{kb}"] = {"nal_codes": codes, "md5": [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in frames]}
for name, which, keep in damage.LOST:  # the slice never comes again: what the reference hands out before the stream is stuck
    frames, codes = ref.decode(damage.truncated_only(name, which, keep))
    out[f"lost-{name}-{which}-{keep}"] = {"nal_codes": codes, "md5": [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in frames]}
for name in damage.DAMAGED_FILES:  # damaged streams kept as files (cases found by tools/damage_sweep.py)
    frames, codes = ref.decode(open(os.path.join(damage.DAMAGED_DIR, name + ".264"), "rb").read())
    out[f"file-{name}"] = {"nal_codes": codes, "md5": [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in frames]}
with open(os.path.join(damage.STREAMS, "damage_md5.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(len(out), "scenarios")
