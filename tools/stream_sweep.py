#!/usr/bin/env python3
"""tools/stream_sweep.py -- CHECKING TOOL (CPU, build container only: needs /root/reference for the syntax writers and the reference decoder): a randomised
differential run of the EMITTERS.  Every seed draws a small Annex-B stream from tests/golden/make_streams.py with all of its options drawn too
(size, GOP shape, CAVLC / CABAC, 8x8 transform, references, weighted prediction, slices, deblocking idc, direct mode, scaling lists, PCM, QPs,
two views, cropping, long-term references / MMCO, list modification, arbitrary slice order, parameter-set switches, frame_num gaps); the stream goes
through the UNMODIFIED reference decoder and through the reference's parser + our emitters + the oracle (capture sink); every NAL's return code and
every frame must agree.  The committed fixtures are 38 fixed points of this space.

    python tools/stream_sweep.py [--seeds A:B]
"""
import argparse
import hashlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_streams as ms  # noqa: E402
from oracle.pyoracle import HipFront, Oracle, ref_decoder  # noqa: E402


DEGENERATE = False
WIDE = False


def options(seed):
    r = random.Random(seed)
    # (no picture one macroblock wide or high: there the REFERENCE's edge-emulation test for 16-sample partitions compares an unsigned position with
    # `size - 16 + 1 - 5` < 0 (src/edge264_inter.c:1199-1204), skips the emulation and predicts from whatever lies beside the plane; the oracle and the
    # kernels follow the standard.  --degenerate draws those sizes too, to look at exactly that.)
    W, H = r.choice([2, 3, 4, 5, 6, 7]), r.choice([2, 3, 4, 5])
    if DEGENERATE:
        W, H = r.choice([1, 2, 3, 4, 5, 6, 7]), r.choice([1, 2, 3, 4, 5])
    frames = r.choice(["IPPP", "IPBPB", "IPBPBB", "II", "IPPBPB", "IPPPBPP", "IPB", "IPP", "IPPB", "IPPPPPP"])
    n = W * H
    o = dict(num_refs=r.randint(1, 4), t8x8=r.random() < 0.4, cabac=r.random() < 0.5, qp=r.randint(12, 46), cqp=(r.randint(-6, 6), r.randint(-6, 6)),
             slices=min(n, r.choice([1, 1, 2, 3, 4])), direct_spatial=r.choice([0, 1]), scaling=r.random() < 0.25,
             pcm=r.choice([0.0, 0.0, 0.03, 0.1]), intra_in_inter=r.choice([0.02, 0.12, 0.4]), skip=r.choice([0.0, 0.15, 0.5]),
             coef_density=r.choice([0.1, 0.35, 0.8]), big_levels=r.choice([0.0, 0.03, 0.2]), cbp_zero=r.choice([0.0, 0.0, 0.5]),
             weighted_pred=r.choice([0, 0, 1]), weighted_bipred=r.choice([0, 1, 2]))
    if "B" in frames:
        o["num_refs"] = max(o["num_refs"], 2)
    o["deblock"] = tuple(sorted(set(r.choice([0, 0, 1, 2]) for _ in range(r.randint(1, 3)))))
    extra = r.random()
    if extra < 0.12:
        o.update(mvc=True, pcm=0.0 if o["cabac"] else o["pcm"])
    elif extra < 0.22:
        o.update(crop=(r.choice([0, 2]), r.choice([0, 4]), r.choice([0, 2]), r.choice([0, 6])))
    elif extra < 0.32:
        o.update(longterm=True, mmco_at=tuple(sorted(r.sample(range(2, 6), r.randint(0, 2)))), num_refs=max(o["num_refs"], 3))
    elif extra < 0.42:
        o.update(reorder=r.choice([0.5, 0.9]), num_refs=max(o["num_refs"], 3))
    elif extra < 0.5 and o["slices"] > 1:
        o.update(aso=True)
    elif extra < 0.58:
        o.update(pps_switch=True)
    elif extra < 0.66 and "B" not in frames:
        o.update(gap_at=tuple(sorted(r.sample(range(2, 5), r.randint(1, 2)))), num_refs=max(o["num_refs"], 2))
    if o["cabac"]:
        o["pcm"] = o["pcm"] if r.random() < 0.5 else 0.0
    if WIDE:  # --wide: the corners of the option space and combinations of the decoder-state features (some the writer refuses: counted)
        W, H = r.choice([2, 3, 5, 8, 12]), r.choice([2, 3, 6, 8])
        n = W * H
        frames = r.choice([frames, "IPPPPPPPPP", "IPBBPBBPBB", "IIII", "IPIPIP", "IPBPBPBPB"])
        o.update(slices=min(n, r.choice([1, 2, 5, 9, n])), intra_in_inter=r.choice([0.0, 0.12, 1.0]), pcm=r.choice([0.0, 0.05, 0.5]),
                 skip=r.choice([0.0, 0.5, 0.95]), cbp_zero=r.choice([0.0, 0.5, 1.0]), qp=r.choice([4, 12, 26, 40, 51]), coef_density=r.choice([0.02, 0.35, 1.0]),
                 big_levels=r.choice([0.0, 0.03, 0.5]))
        if "B" in frames:
            o["num_refs"] = max(o["num_refs"], 2)
            o.pop("gap_at", None)  # (the writer keeps P pictures away from the non-existing frames of a gap, not B pictures: such a stream predicts from them)
        for k, v in (("longterm", True), ("reorder", 0.7), ("pps_switch", True), ("crop", (2, 2, 2, 2))):
            if r.random() < 0.25 and not o.get("mvc"):
                o[k] = v
        if o.get("gap_at"):
            o.pop("reorder", None)  # (a drawn list modification may put the gap's non-existing frame in front: the stream would predict from it)
        if o.get("longterm") or o.get("reorder"):
            o["num_refs"] = max(o["num_refs"], 3)
        if o["slices"] > 1 and r.random() < 0.3:
            o["aso"] = True
    return W, H, frames, o


def md5s(frames):
    return [hashlib.md5(b"".join(p.tobytes() for p in fr)).hexdigest() for fr in frames]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:100")
    ap.add_argument("--degenerate", action="store_true", help="also pictures one macroblock wide or high (a known defect of the reference shows)")
    ap.add_argument("--wide", action="store_true", help="corners of the option space, larger pictures, combinations of the decoder-state features")
    ap.add_argument("--allocators", action="store_true", help="our side decodes with caller allocators (alloc_cb / free_cb of edge264_alloc)")
    ap.add_argument("--lazy", action="store_true", help="fetch frames only when the decoder answers ENOBUFS (and at the end), on both sides")
    args = ap.parse_args()
    import oracle.pyoracle as po
    po.LAZY_DRAIN = args.lazy
    global DEGENERATE, WIDE
    DEGENERATE, WIDE = args.degenerate, args.wide
    a, b = (int(x) for x in args.seeds.split(":"))
    g = ms.load_gen()
    ref, orc = ref_decoder(), Oracle()
    tables = None
    t0, n_ok, n_pics, refused, rejected, bad = time.time(), 0, 0, 0, 0, []
    for seed in range(a, b):
        W, H, frames, o = options(seed)
        if o["cabac"]:
            import cabac_writer as cw
            tables = tables or cw.load_tables()
            o = dict(o, tables=tables)
        try:
            data = ms.Synth(g, f"sweep{seed}", W, H, frames, seed, **o).build()
        except Exception as e:  # a combination the WRITER does not support
            refused += 1
            continue
        f0, c0 = ref.decode(data)
        if len(f0) != len(frames) or not all(c in (0, 105, 61) for c in c0):
            rejected += 1  # the reference itself does not take the stream as meant (a writer limitation): not a case
            continue
        al = None
        if args.allocators:  # the application's own alloc_cb / free_cb (edge264.h:42-43) on our side: every block given back, same frames
            from oracle.pyoracle import CallerAllocator
            al = CallerAllocator()
        f1, c1, _ = HipFront().decode_capture(data, orc, allocator=al)
        if al is not None and (al.allocs != al.frees or al.live):
            bad.append(seed)
            print(f"ALLOCATOR seed {seed}: {al.allocs} allocations, {al.frees} given back", flush=True)
            continue
        n_pics += len(f0)
        if c0 != c1 or md5s(f0) != md5s(f1):
            shown = {k: v for k, v in o.items() if k != "tables"}
            bad.append(seed)
            print(f"MISMATCH seed {seed}: {W}x{H} {frames} {shown}", flush=True)
        else:
            n_ok += 1
    print(f"stream_sweep seeds {a}:{b}: {n_ok} streams ({n_pics} pictures) identical, {len(bad)} mismatches {bad[:10]}, {refused} writer refusals, "
          f"{rejected} not taken by the reference as meant, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
