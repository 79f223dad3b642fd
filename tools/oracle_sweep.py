#!/usr/bin/env python3
"""tools/oracle_sweep.py -- CHECKING TOOL (CPU; needs oracle/_ref/libe264_refkernels.so, i.e. the build container or its prebuilt library): the ORACLE against
the reference's own static kernels (src/edge264_{intra,inter,residual,deblock}.c behind oracle/ref_kernels_harness.c) on synthetic packets whose every
generator option is drawn from the seed -- what no bitstream writer here can produce is in reach this way: vectors hundreds of samples outside the picture,
16 references, explicit weights at the ends of their ranges, QP 0..51, every transform / scaling-list / slice / deblocking combination.  Compared after the
reconstruction pass and after the deblocking pass.  tests/test_oracle_vs_refkernels.py holds the fixed cases.

    python tools/oracle_sweep.py [--seeds A:B]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edge264_amd import packet as P, synth  # noqa: E402
from oracle.pyoracle import Oracle, RefKernels  # noqa: E402


def options(seed):
    r = np.random.default_rng(seed)
    # (no picture one macroblock wide or high: the reference's edge-emulation test underflows there, profiles/r05_sweeps.txt item 1 (ii))
    w, h = int(r.choice([2, 3, 4, 5, 6, 9, 12])), int(r.choice([2, 3, 4, 5, 7]))
    gop = str(r.choice(["IPP", "IPBB", "IPBPB", "IPPBP", "II", "IPBBPBB", "IPPPPPPPB"]))
    nref = int(r.choice([1, 2, 3, 4, 8, 16]))
    kw = dict(num_refs=nref, n_slots=max(6, nref + 2), weighted=int(r.integers(0, 3)), t8x8=bool(r.random() < 0.5), scaling=bool(r.random() < 0.4),
              residual_prob=float(r.choice([0.0, 0.3, 0.8, 1.0])), p_skip=float(r.choice([0.0, 0.1, 0.5])), pcm_prob=float(r.choice([0.0, 0.0, 0.1, 0.4])),
              intra_in_inter=float(r.choice([0.0, 0.05, 0.4])), slices_per_frame=int(r.choice([1, 1, 2, 4, 6])), qp_base=int(r.integers(0, 52)),
              mv_range=int(r.choice([4, 64, 200, 400])), stress=bool(r.random() < 0.3), filter_offsets=(int(r.integers(-12, 13)), int(r.integers(-12, 13))),
              deblock_idc=int(r.choice([0, 0, 1, 2])), deblock=bool(r.random() < 0.9), cabac_like=bool(r.random() < 0.5))
    if kw["t8x8"] and r.random() < 0.5:
        kw["i_kinds"] = (P.MB_I8x8, P.MB_I4x4, P.MB_I16x16)
    kw["slices_per_frame"] = min(kw["slices_per_frame"], w * h)
    return w, h, gop, kw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:300")
    args = ap.parse_args()
    a, b = (int(x) for x in args.seeds.split(":"))
    orc, refk = Oracle(), RefKernels()
    t0, pics, bad, refused = time.time(), 0, [], 0
    for seed in range(a, b):
        w, h, gop, kw = options(seed)
        try:
            s = synth.StreamSynth(w, h, seed, **kw)
        except Exception:
            refused += 1
            continue
        nb = P.frame_bytes(w, h) + 16
        rng = np.random.default_rng(seed + 1000)
        ns = kw["n_slots"]
        dpb_o = [rng.integers(0, 256, nb, dtype=np.uint8) for _ in range(ns)] + [None] * (32 - ns)
        dpb_r = [x.copy() if x is not None else None for x in dpb_o]
        try:
            for i, t in enumerate(gop):
                pkt = s.next_frame(t)
                d = int(P.Packet(pkt).hdr["dst_slot"])
                for passes in (1, 2):
                    orc.decode_frame(pkt, dpb_o, passes)
                    refk.replay(pkt, dpb_r, w, h, passes)
                    if not np.array_equal(dpb_o[d], dpb_r[d]):
                        where = np.flatnonzero(dpb_o[d] != dpb_r[d])[:4].tolist()
                        bad.append(seed)
                        print(f"MISMATCH seed {seed} picture {i} ({t}) pass {passes} {w}x{h} offsets {where} options {kw}", flush=True)
                        dpb_r[d][:] = dpb_o[d]
                pics += 1
        except RuntimeError as e:  # the generator's own limits (DPB too small for the drawn GOP)
            refused += 1
    print(f"oracle_sweep seeds {a}:{b}: {pics} pictures, {len(set(bad))} seeds with a mismatch {sorted(set(bad))[:10]}, {refused} generator refusals, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
