#!/usr/bin/env python3
"""tools/make_gpu_corpus.py -- CHECKING TOOL (build container): writes a corpus of sweep streams (tools/stream_sweep.py narrow + --wide, tools/damage_sweep.py,
tools/nat_sweep.py) with what the unmodified reference decoder makes of each (return codes, md5 of every frame) into ONE file, tools/_gpu_corpus.bin
(git-ignored; it travels with the gpurun snapshot).  tools/check_gpu_corpus.py then runs them through the HIP sink on the device: the sweeps themselves
stop at the oracle, this carries a sample of them to the kernels.
    python tools/make_gpu_corpus.py [--narrow N] [--wide N] [--damage N] [--nat N]"""
import argparse
import contextlib
import hashlib
import io
import json
import os
import pickle
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_streams as ms  # noqa: E402
import stream_sweep as ss  # noqa: E402
import damage_sweep as ds  # noqa: E402
import nat_sweep as ns  # noqa: E402
import nat_encoder as ne  # noqa: E402
from oracle.pyoracle import ref_decoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--narrow", type=int, default=500)
    ap.add_argument("--wide", type=int, default=400)
    ap.add_argument("--damage", type=int, default=300)
    ap.add_argument("--nat", type=int, default=100)
    ap.add_argument("--zip", default=None, help="write the corpus as a zip archive instead (index.json + NNNNN.264, deflated): the committed sample "
                                                "tests/golden/corpus/gpu_corpus_sample.zip that tests/test_gpu_corpus.py runs on the device")
    ap.add_argument("--seed-shift", type=int, default=0, help="added to every seed range: a sample the sweeps and earlier corpora have not seen")
    args = ap.parse_args()
    g = ms.load_gen()
    ref = ref_decoder()
    import cabac_writer as cw
    tables = cw.load_tables()
    md5 = lambda fr: [hashlib.md5(b"".join(p.tobytes() for p in f)).hexdigest() for f in fr]  # noqa: E731
    corpus = []

    def add(kind, seed, data, guarded=False):
        if not guarded:
            f, c = ref.decode(data)
            corpus.append(dict(kind=kind, seed=seed, data=data, codes=c, md5=md5(f)))
            return True
        # a damaged stream may stop the reference at one of its own assertions: decode it in a forked child, take the answer through a pipe
        rd, wr = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(rd)
            f, c = ref.decode(data)
            os.write(wr, pickle.dumps((c, md5(f))))
            os._exit(0)
        os.close(wr)
        buf = b""
        while True:
            chunk = os.read(rd, 65536)
            if not chunk:
                break
            buf += chunk
        os.close(rd)
        _, status = os.waitpid(pid, 0)
        if status != 0 or not buf:
            return False
        c, m = pickle.loads(buf)
        corpus.append(dict(kind=kind, seed=seed, data=data, codes=c, md5=m))
        return True
    for wide, n in ((False, args.narrow), (True, args.wide)):
        ss.WIDE = wide
        for seed in range(700000 + args.seed_shift, 700000 + args.seed_shift + n):
            W, H, frames, o = ss.options(seed)
            if o["cabac"]:
                o = dict(o, tables=tables)
            try:
                data = ms.Synth(g, "c", W, H, frames, seed, **o).build()
            except Exception:
                continue
            add("wide" if wide else "narrow", seed, data)
    ss.WIDE = False
    # damaged streams: only cases the reference survives (each in a child process in the sweep; here: the ones the sweep has already run, seeds 0:..)
    done = 0
    for seed in range(args.seed_shift, args.seed_shift + 10 * args.damage):
        if done >= args.damage:
            break
        W, H, frames, o = ss.options(seed)
        r = random.Random(seed ^ 0x5eed)
        o["slices"] = min(W * H, r.choice([1, 2, 3, 3, 4]))
        if o["slices"] == 1:
            o.pop("aso", None)
        o.pop("mvc", None)
        if o["cabac"]:
            o = dict(o, tables=tables)
        data = ms.Synth(g, "d", W, H, frames, seed, **o).build()
        nals = ds.nal_units(data)
        sl = [i for i, n in enumerate(nals) if (n[3] & 31) in (1, 5)]
        resend = r.random() < 0.67
        if not resend:
            continue  # (lost slices in mid-stream may stop the reference: keep to cut-and-resent here)
        k = r.choice(sl)
        bad = nals[k][:max(6, int(len(nals[k]) * r.uniform(0.15, 0.95)))]
        dmg = b"".join(nals[:k] + [bad] + nals[k:])
        if add("damaged", seed, dmg, guarded=True):
            done += 1
    for seed in range(900000 + args.seed_shift, 900000 + args.seed_shift + args.nat):
        frames, o = ns.options(seed)
        if o["cabac"]:
            o = dict(o, tables=tables)
        with contextlib.redirect_stdout(io.StringIO()):
            data = ne.NatEncoder(g, "n", frames, **o).build(ref)
        add("nat", seed, data)
    out = os.path.join(ROOT, "tools", "_gpu_corpus.bin")
    if args.zip:
        import zipfile
        out = args.zip
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        with zipfile.ZipFile(out, "w", zipfile.ZIP_DEFLATED, compresslevel=9) as z:
            z.writestr("index.json", json.dumps([dict(kind=c["kind"], seed=c["seed"], codes=c["codes"], md5=c["md5"], file=f"{i:05d}.264") for i, c in enumerate(corpus)]))
            for i, c in enumerate(corpus):
                z.writestr(f"{i:05d}.264", c["data"])
    else:
        with open(out, "wb") as f:
            pickle.dump(corpus, f)
    kinds = {}
    for c in corpus:
        kinds[c["kind"]] = kinds.get(c["kind"], 0) + 1
    print(json.dumps(dict(streams=len(corpus), bytes=os.path.getsize(out), kinds=kinds, pictures=sum(len(c["md5"]) for c in corpus))))


# seeds at which the sweep's child stopped at the reference's assertion (cut-and-resent cases among seeds 0:3000 of the first damage sweep)
STOPS = set()

if __name__ == "__main__":
    log = os.path.join(ROOT, "gpurun_out", "sweep", "damage2_0_40000.log")
    if os.path.exists(log):
        import re
        for ln in open(log):
            m = re.match(r"\s*stopped: \((\d+),", ln)
            if m:
                STOPS.add(int(m.group(1)))
    sys.exit(main())
