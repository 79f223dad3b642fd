#!/usr/bin/env python3
"""tools/emu_sweep.py -- CHECKING TOOL (CPU): a randomised differential run of the four kernels' own source (tests/emu: the host build of
edge264_amd/csrc/e264_{dbkp,pred,intra,dbk}.h) against the oracle, whole pipeline per picture -- parameters, prediction + residual, intra, deblocking --
on synthetic streams whose every option is drawn from the seed: picture size (1 x 1 ... 26 x 14 macroblocks), GOP shape, references, weighting
scheme, transforms, scaling lists, QP, residual density, PCM, intra share, slices per picture, far vectors, filter offsets, deblocking idc.
The tests run fixed cases; this is for leaving a few CPU-hours on it.

    python tools/emu_sweep.py [--seeds A:B] [--split 0|1]      prints one line per mismatch and a summary
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edge264_amd import packet as P, synth  # noqa: E402
from oracle.pyoracle import Oracle, _dpb_array  # noqa: E402


def options(seed):
    r = np.random.default_rng(seed)
    w, h = int(r.choice([1, 2, 3, 4, 5, 7, 9, 12, 16, 17, 21, 26])), int(r.choice([1, 2, 3, 4, 5, 6, 8, 9, 11, 14]))
    gop = str(r.choice(["IPP", "IPBB", "IPBPB", "IPPBP", "II", "IPBBPBB", "IBP"]))
    kw = dict(num_refs=int(r.integers(1, 5)), weighted=int(r.integers(0, 3)), t8x8=bool(r.random() < 0.5), scaling=bool(r.random() < 0.3),
              residual_prob=float(r.choice([0.0, 0.1, 0.3, 0.6, 0.9, 1.0])), p_skip=float(r.choice([0.0, 0.1, 0.4, 0.8])),
              pcm_prob=float(r.choice([0.0, 0.0, 0.05, 0.3])), intra_in_inter=float(r.choice([0.0, 0.05, 0.3, 0.7])),
              slices_per_frame=int(r.choice([1, 1, 2, 3, 5])), qp_base=int(r.integers(4, 50)), mv_range=int(r.choice([4, 16, 64, 200, 400])),
              stress=bool(r.random() < 0.25), filter_offsets=(int(r.integers(-6, 7)) * 2 // 2, int(r.integers(-6, 7))), deblock_idc=int(r.choice([0, 0, 1, 2])),
              cabac_like=bool(r.random() < 0.5))
    if r.random() < 0.3:
        kw["i_kinds"] = (P.MB_I4x4, P.MB_I16x16, P.MB_I8x8) if kw["t8x8"] else (P.MB_I4x4,)
    kw["slices_per_frame"] = min(kw["slices_per_frame"], w * h)
    kw["n_slots"] = max(6, kw["num_refs"] + 3)
    return w, h, gop, kw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:200")
    ap.add_argument("--split", type=int, default=1)
    args = ap.parse_args()
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.DEVNULL)
    pe, ie = C.CDLL(os.path.join(d, "libe264_pred_emu.so")), C.CDLL(os.path.join(d, "libe264_intra_emu.so"))
    pe.e264emu_pred_frame.argtypes = [C.c_char_p, C.c_void_p]
    ie.e264emu_intra_frame.argtypes = [C.c_char_p, C.c_void_p]
    pe.e264emu_dbkparam_frame.argtypes = [C.c_char_p, C.c_void_p]
    pe.e264emu_deblock_frame2.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
    a, b = (int(x) for x in args.seeds.split(":"))
    t0, frames, mbs, bad, skipped = time.time(), 0, 0, [], 0
    for seed in range(a, b):
        w, h, gop, kw = options(seed)
        try:
            g = synth.StreamSynth(w, h, seed=seed, **kw)
        except Exception as e:  # an option combination the generator refuses
            skipped += 1
            continue
        nb = P.frame_bytes(w, h)
        rng = np.random.default_rng(seed + 1)
        dpb = [rng.integers(0, 256, nb + 64, dtype=np.uint8) for _ in range(kw["n_slots"])] + [None] * (32 - kw["n_slots"])
        orc = Oracle()
        for k, ft in enumerate(gop):
            try:
                pkt = g.next_frame(ft)
            except Exception:
                skipped += 1
                break
            dst = int(P.Packet(pkt).hdr["dst_slot"])
            mine = [None if x is None else x.copy() for x in dpb]
            orc.decode_frame(pkt, dpb, 3)
            prm = np.zeros(146 * w * h + 64, np.uint8)  # E264_SCRATCH_BYTES
            ok = (pe.e264emu_dbkparam_frame(pkt, prm.ctypes.data) == 0 and pe.e264emu_pred_frame(pkt, _dpb_array(mine)) == 0 and
                  ie.e264emu_intra_frame(pkt, _dpb_array(mine)) == 0)
            if ok:
                r = pe.e264emu_deblock_frame2(pkt, _dpb_array(mine), prm.ctypes.data, args.split)
                ok = r in (0, -1)  # -1: the picture asks for no deblocking at all
            frames += 1
            mbs += w * h
            if not ok or not np.array_equal(mine[dst][:nb], dpb[dst][:nb]):
                where = np.flatnonzero(mine[dst][:nb] != dpb[dst][:nb])[:4].tolist()
                bad.append((seed, k, ft, w, h, kw, where))
                print(f"MISMATCH seed {seed} picture {k} ({ft}) {w}x{h} first offsets {where} options {kw}", flush=True)
                dpb[dst][:] = dpb[dst]  # (the oracle's picture stays the reference of what follows)
    dt = time.time() - t0
    print(f"emu_sweep seeds {a}:{b} split {args.split}: {frames} pictures, {mbs} macroblocks, {len(bad)} mismatches, {skipped} generator refusals, {dt:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
