#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
q = """select s.kernel_name, p.name, e.value, d.end - d.start
       from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
       join rocpd_kernel_dispatch d on e.event_id = d.event_id
       join rocpd_info_kernel_symbol s on d.kernel_id = s.id"""
acc = defaultdict(lambda: defaultdict(list))
for k, n, v, dur in db.execute(q):
    acc[k][n].append(v)
for k, cs in acc.items():
    if "rocclr" in k:
        continue
    print(k[:70])
    for n, vals in sorted(cs.items()):
        print(f"   {n:28s} n={len(vals):4d} avg={sum(vals) / len(vals):16.1f} max={max(vals):16.1f}")
