#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (SQLite) result: per-kernel count / avg / min / max / total,
the same numbers `--stats` prints, as text for profiles/.   usage: rocpd_summary.py results.db"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(
        f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        f"sum(d.end-d.start), max(d.grid_size_x), max(d.workgroup_size_x), max(s.arch_vgpr_count), max(s.sgpr_count), "
        f"max(d.group_segment_size) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc"))
    tot = sum(r[5] for r in rows) or 1
    print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'%':>6s} {'grid':>8s} {'wg':>5s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s}")
    for r in rows:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2] / 1e3:9.2f} {r[3] / 1e3:9.2f} {r[4] / 1e3:9.2f} {r[5] / 1e6:9.3f} "
              f"{100 * r[5] / tot:6.2f} {r[6]:8d} {r[7]:5d} {r[8]:5d} {r[9]:5d} {r[10]:6d}")


if __name__ == "__main__":
    main(sys.argv[1])
