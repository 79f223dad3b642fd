#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table
(the same columns as rocprofv3's --stats CSV: calls, total/avg/min/max duration)."""
import sqlite3
import sys


def main(path: str) -> int:
    db = sqlite3.connect(path)
    q = """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
                  max(d.workgroup_size_x), max(d.grid_size_x), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""
    try:
        rows = db.execute(q).fetchall()
    except sqlite3.OperationalError as e:
        cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
        cols2 = [r[1] for r in db.execute("pragma table_info(rocpd_info_kernel_symbol)")]
        print("schema mismatch:", e, cols, cols2)
        return 1
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}  wg grid vgpr sgpr lds scratch")
    for r in rows:
        print(f"{r[0][:60]:60s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.1f} {r[4] / 1e3:10.1f} {r[5] / 1e3:10.1f} {100 * r[2] / total:6.1f}  {r[6]} {r[7]} {r[8]} {r[9]} {r[10]} {r[11]}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
