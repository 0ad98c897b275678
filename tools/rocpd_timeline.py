#!/usr/bin/env python3
"""Dev aid: the last N kernel dispatches and memory copies of a rocprofv3 rocpd result as a timeline (us since the first of them).
usage: rocpd_timeline.py results.db [N]"""
import sqlite3
import sys


def main(path, n):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    ev = [(s, e, name[:40]) for s, e, name in db.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id")]
    mc = [t for t in tabs if t.startswith("rocpd_memory_copy")]
    if mc:
        cols = [r[1] for r in db.execute(f"pragma table_info({mc[0]})")]
        size = "size" if "size" in cols else cols[-1]
        ev += [(s, e, f"copy {sz} B") for s, e, sz in db.execute(f"select start, end, {size} from {mc[0]}")]
    ev.sort()
    ev = ev[-n:]
    t0 = ev[0][0]
    for s, e, name in ev:
        print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f}  {(e - s) / 1e3:8.1f} us  {name}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24)
