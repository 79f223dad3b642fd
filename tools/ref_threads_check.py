#!/usr/bin/env python3
"""What the UNMODIFIED reference decoder does with n_threads > 0 (edge264_alloc(4, ...), its per-slice worker threads,
/root/reference/src/edge264.c:222-257, src/edge264_headers.c:450-603) on the multi-slice fixtures: every run in a process of its own
with a time limit, output compared with the same library's single-threaded output (= the committed md5s).  This is the evidence behind
DESIGN.md section 7.2 ("n_threads is accepted as a hint"): VERDICT r3 item 7 asked for the worker mode on aso_slices / slices_deblock_idc /
cabac_t8x8_slices bit-exact with edge264_alloc(4, ...); the reference itself is neither bit-exact with its own single-threaded mode there
nor free of hangs.  usage: tools/ref_threads_check.py [runs per stream]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, json, sys
sys.path.insert(0, %r)
from oracle.pyoracle import ref_decoder
fr, codes = ref_decoder().decode(open(sys.argv[1], "rb").read(), n_threads=int(sys.argv[2]))
print(json.dumps([hashlib.md5(b"".join(p.tobytes() for p in f)).hexdigest() for f in fr]))
''' % ROOT


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    sums = json.load(open(os.path.join(ROOT, "tests", "golden", "streams", "reference_md5.json")))
    for name in ["aso_slices", "slices_deblock_idc", "cabac_t8x8_slices", "cabac_slices_deblock_idc", "ipb_spatial", "hd1080_ippb"]:
        path = os.path.join(ROOT, "tests", "golden", "streams", name + ".264")
        res = {"same": 0, "different": 0, "hung (> 20 s)": 0, "crashed": 0}
        for _ in range(runs):
            try:
                p = subprocess.run([sys.executable, "-c", CHILD, path, "4"], capture_output=True, text=True, timeout=20)
                if p.returncode:
                    res["crashed"] += 1
                elif json.loads(p.stdout.strip().splitlines()[-1]) == sums[name]["md5"]:
                    res["same"] += 1
                else:
                    res["different"] += 1
            except subprocess.TimeoutExpired:
                res["hung (> 20 s)"] += 1
        print(f"{name:28s} edge264_alloc(4): {runs} runs -> {res}")


if __name__ == "__main__":
    main()
