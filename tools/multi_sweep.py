#!/usr/bin/env python3
"""tools/multi_sweep.py -- CHECKING TOOL (CPU, build container): the many-decoder driver's host side on random streams.  Batches of 40 streams from
tools/stream_sweep.py's generator (narrow and --wide alternating) are written to files and decoded (a) one decoder at a time through the capture sink and
(b) all at once by edge264_amd/e264_multi --parse-only with 5 threads (any thread any decoder, --stay --ahead 5): the driver must hand out the same number
of frames and its packets must describe the same pictures (sizes, macroblock counts, payloads; tests/test_multi_stream_cpu.py holds one fixed batch).

    python tools/multi_sweep.py [--batches N]
"""
import argparse
import collections
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_streams as ms  # noqa: E402
import stream_sweep as ss  # noqa: E402
from edge264_amd import backend, front, packet as P  # noqa: E402


def shape(p):
    h = P.Packet(p).hdr
    return tuple(int(h[k]) for k in ("total_bytes", "width_mbs", "height_mbs", "n_slices", "n_coded_mbs", "n_inter_mbs", "payload_bytes"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=10)
    args = ap.parse_args()
    g = ms.load_gen()
    import cabac_writer as cw
    tables = cw.load_tables()
    exe, fl = os.path.join(ROOT, "edge264_amd", "e264_multi"), os.path.join(ROOT, "edge264_amd", "libedge264_hipfront.so")
    t0, bad, n_streams, n_frames = time.time(), [], 0, 0
    for bno in range(args.batches):
        ss.WIDE = bool(bno & 1)
        with tempfile.TemporaryDirectory() as tmp:
            files, want, frames = [], collections.Counter(), 0
            for seed in range(400000 + 40 * bno, 400000 + 40 * bno + 40):
                W, H, fr, o = ss.options(seed)
                if o["cabac"]:
                    o = dict(o, tables=tables)
                try:
                    data = ms.Synth(g, "m", W, H, fr, seed, **o).build()
                except Exception:
                    continue
                pk, nf, _ = front.capture_packets(data)
                frames += nf
                for p in pk:
                    want[shape(p)] += 1
                f = os.path.join(tmp, f"s{seed}.264")
                open(f, "wb").write(data)
                files.append(f)
            dump = os.path.join(tmp, "p.e264")
            out = subprocess.run([exe, "--front", fl, "--hip", "/nonexistent", "--parse-only", "--threads", "5", "--stay", "--ahead", "5", "--dump-packets", dump] + files,
                                 capture_output=True, text=True, timeout=600)
            ok = out.returncode == 0
            got = collections.Counter()
            if ok:
                st = json.loads(out.stdout.strip().splitlines()[-1])
                raw, off = open(dump, "rb").read(), 0
                while off < len(raw):
                    n = int.from_bytes(raw[off + 8:off + 12], "little")
                    pkt = raw[off:off + n]
                    ok = ok and backend.packet_check(pkt) == 0
                    got[shape(pkt)] += 1
                    off += n
                ok = ok and st["frames"] == frames and got == want and st["stuck_decoders_flushed"] == 0
            n_streams += len(files)
            n_frames += frames
            if not ok:
                bad.append(bno)
                print(f"MISMATCH batch {bno}: exit {out.returncode}, frames {frames}, {out.stdout.strip()[-300:]} {out.stderr.strip()[-300:]}", flush=True)
    print(f"multi_sweep: {args.batches} batches, {n_streams} decoders, {n_frames} frames: {len(bad)} batches differ {bad}, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
