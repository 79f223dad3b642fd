#!/usr/bin/env python3
"""Per-kernel PMC counters of a rocprofv3 rocpd database: a dispatch's counter is reported as
several rows (one per XCD / instance); they are SUMMED per dispatch, then averaged over dispatches.

    pmc_summary.py results.db [more.db ...] [--traffic out.json --streams N --gop G --width-mbs W --height-mbs H]

--traffic writes the HBM bytes per launch of every kernel: FETCH_SIZE (KB; doubled: on gfx950 the
counter tallies 128-B requests at 64 B, /opt/skills/guides/MI355X_MICROARCH.md "HBM") + WRITE_SIZE (KB).
"""
import argparse
import re
import json
import sqlite3
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dbs", nargs="+")
ap.add_argument("--traffic")
ap.add_argument("--streams", type=int)
ap.add_argument("--gop")
ap.add_argument("--width-mbs", type=int, default=120)
ap.add_argument("--height-mbs", type=int, default=68)
ap.add_argument("--sq-json", help="write the per-kernel SQ counters (wave-instructions per launch by kind, waves, wave cycles) as JSON: bench.py derives the VALU issue share from the newest profiles/r*_sq_mix.json")
ap.add_argument("--bench-json", help="a bench.py line of the SAME build: its live kernel times are recorded in the traffic file (bench.py flags the file as stale when they drift)")
args = ap.parse_args()

q = """select s.kernel_name, p.name, e.value, d.id
       from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
       join rocpd_kernel_dispatch d on e.event_id = d.event_id
       join rocpd_info_kernel_symbol s on d.kernel_id = s.id"""
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for path in args.dbs:
    db = sqlite3.connect(path)
    for k, n, v, did in db.execute(q):
        acc[k][n][(path, did)] += v
kern = {}
# "per launch" = per SUBMISSION of the GOP: a kernel that a submission leaves out (e264_pred_kernel on all-intra batches since round 6) counts as zero there, exactly as
# bench.py's live kernel times do (its ms_per_launch divides by the number of submissions)
n_sub = max((len(per) for k, cs in acc.items() if "rocclr" not in k for per in cs.values()), default=1)
for k, cs in acc.items():
    if "rocclr" in k:
        continue
    m = re.search(r"(e264_[a-z0-9]+_kernel|k_[a-z0-9_]+?)(?=P|I|$|\.)", k)  # mangled (_Z17e264_mbpar_kernelPK...) or plain names
    short = m.group(1) if m else k.split("(")[0].split("<")[0].replace("void ", "")
    print(k[:90])
    for n, per in sorted(cs.items()):
        vals = list(per.values())
        avg = sum(vals) / n_sub
        print(f"   {n:28s} dispatches={len(vals):4d} of {n_sub} submissions  avg per submission={avg:18.1f} max={max(vals):18.1f}")
        kern.setdefault(short, {})[n] = avg
if args.traffic:
    out = {"config": {"streams": args.streams, "gop": args.gop, "width_mbs": args.width_mbs, "height_mbs": args.height_mbs},
           "note": "HBM bytes per launch = 2 x FETCH_SIZE(KB) x 1024 (gfx950 correction) + WRITE_SIZE(KB) x 1024, averaged over all launches of the GOP",
           "kernels": {}}
    for k, c in kern.items():
        if "TCC_EA0_RDREQ_128B" in c and "TCC_EA0_WRREQ" in c:
            # raw request counters by size (gfx950 lists them; FETCH_SIZE's gfx94x formula mis-sizes 128-B requests)
            rd = 32 * c["TCC_EA0_RDREQ_32B"] + 64 * c["TCC_EA0_RDREQ_64B"] + 128 * c["TCC_EA0_RDREQ_128B"]
            wr = 64 * c["TCC_EA0_WRREQ_64B"] + 32 * (c["TCC_EA0_WRREQ"] - c["TCC_EA0_WRREQ_64B"])
            out["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                                 "rdreq": {n: c[n] for n in c if "RDREQ" in n}, "wrreq": {n: c[n] for n in c if "WRREQ" in n}}
        elif "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            out["kernels"][k] = {"fetch_kb": c["FETCH_SIZE"], "write_kb": c["WRITE_SIZE"],
                                 "hbm_bytes_per_launch": 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024}
    out["note"] = ("HBM bytes per launch from the L2 memory-side request counters, sized individually: reads 32*RDREQ_32B + 64*RDREQ_64B + "
                   "128*RDREQ_128B, writes 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B); calibrated on tools/calib/pmc_calib.hip "
                   "(profiles/r01_pmc_calibration.txt); averaged over all launches of the GOP")
    if args.bench_json:
        with open(args.bench_json) as f:
            bj = json.load(f)
        out["kernel_ms_per_launch"] = {k: v["ms_per_launch"] for k, v in bj["roofline"]["kernels"].items()}
        out["taken_with"] = {"value": bj["value"], "build_flags": bj.get("build_flags")}
    with open(args.traffic, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.traffic)

if args.sq_json:
    out = {"config": {"streams": args.streams, "gop": args.gop, "width_mbs": args.width_mbs, "height_mbs": args.height_mbs},
           "note": "SQ counters per launch (summed over XCDs / instances, averaged over all launches of the GOP): wave-level instruction counts by kind",
           "kernels": {k: {n: v for n, v in c.items() if n.startswith("SQ_")} for k, c in kern.items() if any(n.startswith("SQ_") for n in c)}}
    if args.bench_json:
        with open(args.bench_json) as f:
            bj = json.load(f)
        out["kernel_ms_per_launch"] = {k: v["ms_per_launch"] for k, v in bj["roofline"]["kernels"].items()}
    with open(args.sq_json, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.sq_json)
