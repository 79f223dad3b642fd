#!/usr/bin/env python3
"""tools/damage_sweep.py -- CHECKING TOOL (CPU, build container only): the concealment path of the emitters against the unmodified reference, randomised.
Every seed: a small random stream (tools/stream_sweep.py's generator, several slices per picture more often than not), one of its slice NALs cut short
at a random place and -- two times out of three -- sent again intact behind the damaged copy, sometimes with the next slice of the same picture cut too
before both are sent again (tests/damage.py holds 75 fixed scenarios of these kinds).
Both decoders see the same bytes; every NAL's return code and every frame handed out must agree.  Cases run in child processes, 40 at a time: a damaged
stream may stop at one of the reference's own assertions (src/edge264_headers.c:465), which ends the child, not the run.

    python tools/damage_sweep.py [--seeds A:B]
"""
import argparse
import hashlib
import json
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def nal_units(data):
    pos, i = [], 0
    while True:
        j = data.find(b"\0\0\1", i)
        if j < 0:
            break
        pos.append(j)
        i = j + 3
    return [data[a:b] for a, b in zip(pos, pos[1:] + [len(data)])]


MORE = False
NAT = False
_REF = [None]


def make_nat_case(seed, g, tables_box):
    """--nat: an encoder-shaped clip (tools/nat_sweep.py's options; the reference closes the encoder's loop) with one slice NAL cut and sent again"""
    import contextlib
    import io
    import nat_encoder as ne
    import nat_sweep as ns
    from oracle.pyoracle import ref_decoder
    frames, o = ns.options(seed)
    r = random.Random(seed ^ 0x5eed)
    if r.random() < 0.7 and "slice_rows" not in o:
        o["slice_rows"] = r.randint(1, max(1, o["H"] - 1))  # several slices per picture more often than not
    if o["cabac"]:
        import cabac_writer as cw
        tables_box[0] = tables_box[0] or cw.load_tables()
        o = dict(o, tables=tables_box[0])
    _REF[0] = _REF[0] or ref_decoder()
    info = dict(size=f"{o['W']}x{o['H']}", gop=frames, slices=-(-o["H"] // o["slice_rows"]) if o.get("slice_rows") else 1, options={k: v for k, v in o.items() if k != "tables"})
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            data = ne.NatEncoder(g, f"n{seed}", frames, **o).build(_REF[0])
    except Exception:
        return None, info
    nals = nal_units(data)
    sl = [i for i, n in enumerate(nals) if (n[3] & 31) in (1, 5)]
    resend = r.random() < 0.75
    k = r.choice(sl) if resend else r.choice(sl[-info["slices"]:])
    bad = nals[k][:max(6, int(len(nals[k]) * r.uniform(0.15, 0.95)))]
    dmg = b"".join(nals[:k] + [bad] + (nals[k:] if resend else nals[k + 1:]))
    info.update(resend=resend, two=False, slice=k, cut_to=len(bad), of=len(nals[k]))
    return dmg, info


def make_case(seed, g, tables_box):
    """-> (damaged stream, info) of one seed, or (None, info) when the writer refuses the drawn options"""
    import make_streams as ms
    import stream_sweep as ss
    if NAT:
        return make_nat_case(seed, g, tables_box)
    W, H, frames, o = ss.options(seed)
    r = random.Random(seed ^ 0x5eed)
    _pick = r.choice([1, 2, 3, 3, 4])  # (drawn in both modes: the same seed stays the same case)
    if not ss.WIDE:
        o["slices"] = min(W * H, _pick)
    o.pop("aso", None) if o["slices"] == 1 else None
    o.pop("mvc", None)  # (a failed slice of the second view: the reference's own territory of assertions)
    if o["cabac"]:
        import cabac_writer as cw
        tables_box[0] = tables_box[0] or cw.load_tables()
        o = dict(o, tables=tables_box[0])
    info = dict(size=f"{W}x{H}", gop=frames, slices=o["slices"], options={k: v for k, v in o.items() if k != "tables"})
    try:
        data = ms.Synth(g, f"d{seed}", W, H, frames, seed, **o).build()
    except Exception:
        return None, info
    nals = nal_units(data)
    sl = [i for i, n in enumerate(nals) if (n[3] & 31) in (1, 5)]
    resend = r.random() < 0.67
    # (a slice that never comes again leaves its picture incomplete: only in the LAST picture, or the reference stops at its assertion about
    # incomplete reference frames as soon as a later picture predicts from it)
    k = r.choice(sl) if resend else r.choice(sl[-o["slices"]:])
    bad = nals[k][:max(6, int(len(nals[k]) * r.uniform(0.15, 0.95)))]
    dmg = b"".join(nals[:k] + [bad] + (nals[k:] if resend else nals[k + 1:]))
    two = False
    if resend and o["slices"] >= 2 and not o.get("aso") and r.random() < 0.3:
        # two failures inside ONE picture before anything is sent again (pairs that span two pictures stop the reference at its assertion)
        j = sl.index(k)
        if j + 1 < len(sl) and j // o["slices"] == (j + 1) // o["slices"]:
            k2 = sl[j + 1]
            bad2 = nals[k2][:max(6, int(len(nals[k2]) * r.uniform(0.15, 0.95)))]
            dmg = b"".join(nals[:k] + [bad, bad2, nals[k], nals[k2]] + nals[k2 + 1:])
            two = True
    if MORE and resend and not two and r.random() < 0.5:
        # --more: further slices of the stream cut and sent again, each inside its own picture (every failure is repaired before the next one)
        out, n_cut = [], 0
        for i, n in enumerate(nals):
            if i in sl and (i == k or r.random() < 0.25):
                frac = r.choice([0.0, 0.02, 0.999]) if r.random() < 0.3 else r.uniform(0.1, 0.97)  # also: the header alone, nearly nothing, nearly everything
                out.append(n[:max(5, int(len(n) * frac))])
                n_cut += 1
            out.append(n)
        dmg = b"".join(out)
        info["cuts"] = n_cut
    info.update(resend=resend, two=two, slice=k, cut_to=len(bad), of=len(nals[k]))
    return dmg, info


def child(a, b):
    import make_streams as ms
    from oracle.pyoracle import HipFront, Oracle, ref_decoder
    g = ms.load_gen()
    ref, orc = ref_decoder(), Oracle()
    box = [None]
    md5 = lambda fr: [hashlib.md5(b"".join(p.tobytes() for p in f)).hexdigest() for f in fr]  # noqa: E731
    for seed in range(a, b):
        dmg, info = make_case(seed, g, box)
        if dmg is None:
            print(json.dumps({"seed": seed, "status": "refused"}), flush=True)
            continue
        print(json.dumps({"seed": seed, "status": "start"}), flush=True)
        f0, c0 = ref.decode(dmg)
        f1, c1, _ = HipFront().decode_capture(dmg, orc)
        same = c0 == c1 and md5(f0) == md5(f1)
        if not same:
            # is the reference itself deterministic here?  A damaged stream can make it predict from frames it never wrote (the non-existing frames of a
            # frame_num gap in freshly allocated buffers): its output then depends on what the heap held.  Decode again with the heap disturbed.
            junk = [bytes([(i * 7 + seed) % 251]) * (1 << 20) for i in range(60)]
            f2, c2 = ref_decoder().decode(dmg)
            del junk
            if c2 != c0 or md5(f2) != md5(f0):
                print(json.dumps({"seed": seed, "status": "reference_not_deterministic"}), flush=True)
                continue
        print(json.dumps({"seed": seed, "status": "same" if same else "MISMATCH", "frames": len(f0), "resend": info["resend"], "two": info["two"], "slice": info["slice"],
                          "slices": info["slices"], "size": info["size"], "gop": info["gop"], "codes_equal": c0 == c1, "n": (len(f0), len(f1))}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:400")
    ap.add_argument("--child", default=None)
    ap.add_argument("--wide", action="store_true", help="stream_sweep.py --wide's option space (larger pictures, one slice per macroblock, all-intra / all-PCM ...)")
    ap.add_argument("--nat", action="store_true", help="encoder-shaped clips (tools/nat_sweep.py's generator) instead of random syntax")
    ap.add_argument("--more", action="store_true", help="several slices of a stream cut and sent again, cuts at the extremes too")
    ap.add_argument("--lazy", action="store_true", help="fetch frames only when the decoder answers ENOBUFS (and at the end), on both sides")
    args = ap.parse_args()
    import oracle.pyoracle as po
    po.LAZY_DRAIN = args.lazy
    import stream_sweep as ss
    global MORE, NAT
    ss.WIDE, MORE, NAT = args.wide, args.more, args.nat
    if args.child:
        a, b = (int(x) for x in args.child.split(":"))
        child(a, b)
        return 0
    a, b = (int(x) for x in args.seeds.split(":"))
    t0 = time.time()
    tally = dict(same=0, MISMATCH=0, refused=0, reference_stopped=0, reference_not_deterministic=0)
    bad, stopped = [], []
    s = a
    while s < b:
        e = min(b, s + 40)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", f"{s}:{e}"] + (["--lazy"] if args.lazy else []) + (["--wide"] if args.wide else []) + (["--more"] if args.more else []) + (["--nat"] if args.nat else []), capture_output=True, text=True, timeout=1200)
        last_started = None
        for ln in p.stdout.splitlines():
            if not ln.startswith("{"):
                continue
            d = json.loads(ln)
            if d["status"] == "start":
                last_started = d["seed"]
                continue
            last_started = None
            tally[d["status"]] += 1
            if d["status"] == "MISMATCH":
                bad.append(d)
                print(d, flush=True)
        if p.returncode != 0 and last_started is not None:  # the child died inside a case: the reference's assertion (or a crash: stderr says which)
            tally["reference_stopped"] += 1
            stopped.append((last_started, p.stderr.strip().splitlines()[-1][:160] if p.stderr.strip() else f"exit {p.returncode}"))
            s = last_started + 1
            continue
        s = e
    print(f"damage_sweep seeds {a}:{b}: {tally}, {time.time() - t0:.0f} s")
    for st in stopped[:12]:
        print("  stopped:", st)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
