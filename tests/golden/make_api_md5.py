#!/usr/bin/env python3
"""tests/golden/make_api_md5.py -- GENERATING SCRIPT of tests/golden/api/api_md5.json.

tests/golden/api/*.264 are the reference's own API-behaviour fixtures (data files of /root/reference/tests: parameter-set handling, frame
finishing, POC order, supported / unsupported NAL types, cropping), copied as they are.  This script runs the UNMODIFIED reference decoder
(oracle/_ref/libedge264_ref.so, built from /root/reference by oracle/Makefile) over each of them and records what it answers: the return code
of every edge264_decode_NAL call and the md5 of every frame edge264_get_frame hands out -- the convention of streams/reference_md5.json.
tests/test_frontend_hip.py holds the product (libedge264_hipfront.so on the HIP back end) against it on the GPU box, where neither the
reference tree nor anything built at test time from it exists.

    python tests/golden/make_api_md5.py
"""
import glob
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.pyoracle import ref_decoder  # noqa: E402


def main():
    src = "/root/reference/tests"
    out = {}
    ref = ref_decoder()
    for path in sorted(glob.glob(os.path.join(HERE, "api", "*.264"))):
        name = os.path.basename(path)[:-4]
        data = open(path, "rb").read()
        if os.path.isdir(src):
            assert data == open(os.path.join(src, name + ".264"), "rb").read(), f"{name}: not the reference's file"
        frames, codes = ref.decode(data)
        out[name] = {"md5": [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in frames], "nal_codes": codes,
                     "frame_shapes": [list(fr[0].shape) for fr in frames]}
        print(name, len(frames), "frames", codes)
    with open(os.path.join(HERE, "api", "api_md5.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
