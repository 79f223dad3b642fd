#!/usr/bin/env python3
"""tools/sanitize/run.py -- CHECKING TOOL.  Drives the AddressSanitizer + UndefinedBehaviorSanitizer build of the product front end (the
reference's parser with our emitters, capture sink: no device involved) over
  * every fixture of tests/golden/streams,
  * every damaged-stream scenario of tests/damage.py (a slice cut and sent again, two slices of one picture, a slice lost),
  * seeded corruptions: bit flips anywhere behind the parameter sets, bit flips inside the last slice only, truncations at arbitrary bytes,
and sorts what the sanitizers say by WHERE it happened: in the emitters / wrappers of edge264_amd/frontend (ours: must be none), or inside
the reference's own parser sources (reported, not ours to change).  A corrupted stream may also stop at one of the reference's own
assertions (src/edge264_headers.c:465): counted, not an error of the binding.
A second leg runs the many-decoder driver (edge264_amd/driver/e264_multi.cpp, --parse-only: no device) and the front end under ThreadSanitizer.

    python tools/sanitize/run.py [--flips N] [--out profiles/r05_sanitizers.txt]
"""
import argparse
import glob
import os
import random
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import damage  # noqa: E402

OURS = ("e264_emit.h", "emit_deblock.c", "emit_inter.c", "emit_intra.c", "emit_residual.c", "edge264_hip_frontend.c")


def corrupt(data: bytes, seed: int, flips: int) -> bytes:
    rng = random.Random(seed)
    d = bytearray(data)
    first_slice = min((i for i in range(len(d) - 4) if d[i:i + 3] == b"\0\0\1" and (d[i + 3] & 31) in (1, 5, 20)), default=0)
    n = 0
    while n < flips:
        i = rng.randrange(first_slice + 5, len(d) - 1)
        if d[i] in (0, 1) or d[i - 1] == 0 or d[i + 1] == 0:  # never make or break a start code
            continue
        d[i] ^= 1 << rng.randrange(8)
        n += 1
    return bytes(d)


def corrupt_last(data: bytes, seed: int, flips: int) -> bytes:
    """flips inside the LAST slice NAL only: no later picture predicts from it, so the reference's own assertion about incomplete reference
    frames (src/edge264_headers.c:465) stays out of the way and the emitters see the damage"""
    rng = random.Random(seed)
    d = bytearray(data)
    last = bytes(d).rfind(b"\0\0\1")
    n = 0
    while n < flips and len(d) - last > 12:
        i = rng.randrange(last + 8, len(d) - 1)
        if d[i] in (0, 1) or d[i - 1] == 0 or d[i + 1] == 0:
            continue
        d[i] ^= 1 << rng.randrange(8)
        n += 1
    return bytes(d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flips", type=int, default=3, help="corrupted variants per fixture")
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-tsan", action="store_true", help="skip the ThreadSanitizer leg (e264_multi --parse-only)")
    ap.add_argument("--compact", action="store_true", help="the front end folds its packets (E264_FRONT_COMPACT=1: the wire form, include/edge264_compact.h)")
    args = ap.parse_args()
    if args.compact:
        os.environ["E264_FRONT_COMPACT"] = "1"
    subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)
    exe, lib = os.path.join(HERE, "hostprof_san"), os.path.join(HERE, "libedge264_hipfront_san.so")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0")
    cases = []
    fixtures = sorted(glob.glob(os.path.join(damage.STREAMS, "*.264")))
    tmp = tempfile.mkdtemp(prefix="e264san_", dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)

    def put(name, data):
        p = os.path.join(tmp, name + ".264")
        with open(p, "wb") as f:
            f.write(data)
        return p
    for f in fixtures:
        cases.append(("fixture", os.path.basename(f)[:-4], f))
    for n, w, k in damage.RESENT:
        cases.append(("resent", f"{n}-{w}-{k}", put(f"resent-{n}-{w}-{k}", damage.truncated_then_resent(n, w, k))))
    for n, w, ka, kb in damage.RESENT2:
        cases.append(("resent2", f"{n}-{w}-{ka}-{kb}", put(f"resent2-{n}-{w}", damage.two_truncated_then_resent(n, w, ka, kb))))
    for n, w, k in damage.LOST:
        cases.append(("lost", f"{n}-{w}-{k}", put(f"lost-{n}-{w}", damage.truncated_only(n, w, k))))
    for f in fixtures:
        data = open(f, "rb").read()
        if len(data) > 300_000:
            continue
        name = os.path.basename(f)[:-4]
        for s in range(args.flips):
            cases.append(("bitflips", f"{name}-seed{s}", put(f"flip-{name}-{s}", corrupt(data, 1000 + s, 1 + 2 * s))))
            cases.append(("flips_last", f"{name}-seed{s}", put(f"fliplast-{name}-{s}", corrupt_last(data, 2000 + s, 1 + s))))
        rng = random.Random(name)
        for s in range(2):
            cut = rng.randrange(len(data) // 3, len(data))
            cases.append(("truncated", f"{name}-at{cut}", put(f"cut-{name}-{s}", data[:cut])))
    tally = {}
    ours, theirs = {}, {}
    farm_ours = tuple(f"_farm/src/edge264_{k}.c" for k in ("deblock", "inter", "intra", "residual"))  # symlinks to emit_*.c

    def is_ours(frame):
        return any(f in frame for f in farm_ours) or ("/edge264_amd/frontend/" in frame and any(o in frame for o in OURS))
    for kind, name, path in cases:
        p = subprocess.run([exe, lib, "0.001", "-1", path], env=env, capture_output=True, text=True, timeout=600)
        t = tally.setdefault(kind, dict(runs=0, clean=0, reference_assert=0, reports_ours=0, reports_reference=0, other_exit=0))
        t["runs"] += 1
        # one report = its headline + the frames that follow it; it belongs to whoever owns the innermost frame that is not the sanitizer's own
        blocks, cur = [], None
        for ln in p.stderr.splitlines():
            if re.search(r"runtime error: |ERROR: AddressSanitizer|ERROR: LeakSanitizer", ln):
                cur = [re.sub(r"^.*?(runtime error: |ERROR: )", r"\1", ln).strip(), []]
                blocks.append(cur)
            elif cur is not None and re.match(r"\s*#\d+ ", ln):
                cur[1].append(ln.strip())
        if not blocks:
            if p.returncode == 0:
                t["clean"] += 1
            elif "Assertion" in p.stderr:
                t["reference_assert"] += 1
            else:
                t["other_exit"] += 1
                theirs.setdefault((f"exit code {p.returncode}", p.stderr.strip()[-200:]), []).append(f"{kind} {name}")
            continue
        for head, frames in blocks:
            inner = next((f for f in frames if "libsanitizer" not in f and "interceptor" not in f), frames[0] if frames else "")
            where = re.sub(r"^#\d+ 0x[0-9a-f]+ in ", "", inner)
            mine = is_ours(inner)
            (ours if mine else theirs).setdefault((head, where), []).append(f"{kind} {name}")
            t["reports_ours" if mine else "reports_reference"] += 1
    lines = ["AddressSanitizer + UndefinedBehaviorSanitizer over the front end (reference parser + emitters, capture sink), tools/sanitize/run.py", ""]
    for kind, t in tally.items():
        lines.append(f"{kind:10s} " + "  ".join(f"{k} {v}" for k, v in t.items()))
    lines.append("")
    lines.append(f"distinct reports located in the emitters / wrappers (edge264_amd/frontend): {len(ours)}")
    for (head, where), runs in ours.items():
        lines.append(f"  {head}\n      at {where}\n      in {len(runs)} runs, e.g. {runs[0]}")
    lines.append(f"distinct reports located in the reference's own sources (its parser, compiled where it lies), or other exits: {len(theirs)}")
    for (head, where), runs in theirs.items():
        lines.append(f"  {head}\n      at {where}\n      in {len(runs)} runs, e.g. {runs[0]}")
    # ---- ThreadSanitizer: the many-decoder driver, parse-only (no device), any thread advances any decoder ----
    if not args.no_tsan:
        subprocess.run(["make", "-C", HERE, "tsan"], check=True, stdout=subprocess.DEVNULL)
        multi, tlib = os.path.join(HERE, "e264_multi_tsan"), os.path.join(HERE, "libedge264_hipfront_tsan.so")
        small = [f for f in fixtures if os.path.getsize(f) < 60_000]
        damaged = [c[2] for c in cases if c[0] in ("resent", "resent2", "lost")]
        lines += ["", "ThreadSanitizer over e264_multi --parse-only (driver and front end built with it; threads x decoders, --stay --ahead 5, every stream played twice):"]
        tsan_bad = 0
        for label, files, threads in (("intact small fixtures", small, 2), ("intact small fixtures", small, 5), ("intact small fixtures", small, 16),
                                      ("damaged streams", damaged, 4), ("both, no --stay", small + damaged, 7)):
            cmd = [multi, "--front", tlib, "--hip", "/nonexistent", "--parse-only", "--threads", str(threads), "--repeat", "2"]
            cmd += [] if "no --stay" in label else ["--stay", "--ahead", "5"]
            p = subprocess.run(cmd + files, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"), capture_output=True, text=True, timeout=900)
            n = p.stderr.count("WARNING: ThreadSanitizer")
            tsan_bad += n + (p.returncode != 0)
            m = re.search(r'"frames": (\d+)', p.stdout)
            lines.append(f"  {label:24s} {len(files):3d} decoders x2  {threads:2d} threads   frames {m.group(1) if m else '?':>5s}   warnings {n}   exit {p.returncode}")
            for w in re.findall(r"SUMMARY: ThreadSanitizer: [^\n]*", p.stderr)[:5]:
                lines.append(f"      {w}")
        if tsan_bad:
            ours["tsan"] = ["tsan"]
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    return 1 if ours else 0


if __name__ == "__main__":
    sys.exit(main())
