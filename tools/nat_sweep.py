#!/usr/bin/env python3
"""tools/nat_sweep.py -- CHECKING TOOL (CPU, build container only): tools/stream_sweep.py's comparison on ENCODER-SHAPED streams.  Every seed: a short
clip of the procedural scene through tests/golden/nat_encoder.py with its options drawn (size 3..9 x 2..6 macroblocks, I/P/B pattern, CAVLC / CABAC, QP,
High-profile tools, adaptive QP, slices every few rows, deblocking idc, a fade with explicit weights, 16x8 / 8x16 and sub-8x8 partitions, Intra8x8);
the reference decoder closes the encoder's loop, then the finished stream goes through the reference and through its parser + our emitters + the oracle.

    python tools/nat_sweep.py [--seeds A:B]
"""
import argparse
import hashlib
import io
import contextlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_streams as ms  # noqa: E402
import nat_encoder as ne  # noqa: E402
from oracle.pyoracle import HipFront, Oracle, ref_decoder  # noqa: E402


def options(seed):
    r = random.Random(seed)
    W, H = r.choice([3, 4, 5, 7, 9]), r.choice([2, 3, 4, 6])
    high = r.random() < 0.5
    cabac = r.random() < 0.5
    frames = r.choice(["IPPP", "IPBBP", "IPBPB", "IIPP", "IPPPPP", "IPBBPBB"])
    o = dict(cabac=cabac, qp=r.randint(18, 40), seed=seed, W=W, H=H, high=high, search=r.choice([4, 8]))
    if r.random() < 0.4:
        o["aq"] = r.randint(1, 5)
    if r.random() < 0.4:
        o["slice_rows"] = r.randint(1, H)
    if r.random() < 0.3:
        o["deblock_idc"] = r.choice([1, 2])
    if "B" not in frames and r.random() < 0.3:
        o["fade"] = (0, len(frames), r.choice([0.2, 0.5, 1.6]))
    if r.random() < 0.5:
        o["rect"] = True
    if r.random() < 0.4:
        o["sub"] = True
    if high and r.random() < 0.5:
        o["i8x8"] = True
    return frames, o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:50")
    ap.add_argument("--lazy", action="store_true", help="fetch frames only when the decoder answers ENOBUFS (and at the end), on both sides")
    args = ap.parse_args()
    import oracle.pyoracle as po
    po.LAZY_DRAIN = args.lazy
    a, b = (int(x) for x in args.seeds.split(":"))
    g = ms.load_gen()
    ref, orc = ref_decoder(), Oracle()
    tables = None
    md5 = lambda fr: [hashlib.md5(b"".join(p.tobytes() for p in f)).hexdigest() for f in fr]  # noqa: E731
    t0, ok, pics, bad, failed = time.time(), 0, 0, [], 0
    for seed in range(a, b):
        frames, o = options(seed)
        if o["cabac"]:
            import cabac_writer as cw
            tables = tables or cw.load_tables()
            o = dict(o, tables=tables)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                data = ne.NatEncoder(g, f"n{seed}", frames, **o).build(ref)
        except Exception as e:  # the encoder's own loop check (the reference must decode every picture so far) or an option it cannot combine
            failed += 1
            print(f"encoder stopped, seed {seed}: {type(e).__name__} {str(e)[:120]} {dict((k, v) for k, v in o.items() if k != 'tables')}", flush=True)
            continue
        f0, c0 = ref.decode(data)
        f1, c1, _ = HipFront().decode_capture(data, orc)
        pics += len(f0)
        if c0 != c1 or md5(f0) != md5(f1):
            bad.append(seed)
            print(f"MISMATCH seed {seed}: {frames} {dict((k, v) for k, v in o.items() if k != 'tables')}", flush=True)
        else:
            ok += 1
    print(f"nat_sweep seeds {a}:{b}: {ok} streams ({pics} pictures) identical, {len(bad)} mismatches {bad[:10]}, {failed} stopped inside the encoder, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
