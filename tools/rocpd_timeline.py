#!/usr/bin/env python3
"""Per-step GPU timeline from a rocprofv3 kernel trace (rocpd SQLite): for every kernel of the
encode sequence, its average duration and the average idle gap between the end of the previous
dispatch and its start (steady state = the last 60% of the dispatches).  usage: rocpd_timeline.py results.db"""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    rows = rows[int(len(rows) * 0.4):]
    dur, gap, n = defaultdict(float), defaultdict(float), defaultdict(int)
    order = []
    prev_end = None
    for name, st, en in rows:
        short = name.split("(")[0].split("<")[0].split("::")[-1]
        if short not in order:
            order.append(short)
        dur[short] += en - st
        if prev_end is not None:
            gap[short] += st - prev_end
        n[short] += 1
        prev_end = max(prev_end or 0, en)
    print(f"{'kernel':40s} {'calls':>6s} {'avg_us':>9s} {'gap_before_us':>14s}")
    tot = 0.0
    for k in order:
        print(f"{k:40s} {n[k]:6d} {dur[k] / n[k] / 1e3:9.2f} {gap[k] / n[k] / 1e3:14.2f}")
        tot += (dur[k] + gap[k]) / n[k] / 1e3
    print(f"sum of (duration + gap) per step: {tot:.2f} us")


if __name__ == "__main__":
    main(sys.argv[1])
