#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 --pmc result databases (rocpd sqlite).

    python tools/pmc_summary.py gpurun_out/pmc_*/p_results.db [--kernel k_pretok]
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    kern = None
    if "--kernel" in sys.argv:
        kern = sys.argv[sys.argv.index("--kernel") + 1]
        args.remove(kern)
    for path in args:
        db = sqlite3.connect(path)
        cols = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
        cname = "counter_name" if "counter_name" in cols else ("pmc_name" if "pmc_name" in cols else None)
        vname = "counter_value" if "counter_value" in cols else "value"
        if cname is None:
            print(path, "unexpected schema", cols)
            continue
        acc = defaultdict(lambda: [0.0, 0, 0.0])
        ndisp = defaultdict(set)
        for name, cn, val, did, dur in db.execute(f"select name, {cname}, {vname}, dispatch_id, duration from pmc_events"):
            short = name.split("(")[0].split("<")[0].split("::")[-1]
            if kern and kern not in name:
                continue
            a = acc[(short, cn)]
            a[0] += val
            ndisp[(short, cn)].add(did)
            a[2] += dur
        for (short, cn), a in sorted(acc.items()):
            n = len(ndisp[(short, cn)])
            print(f"{short:28s} {cn:28s} per launch {a[0] / n:16.1f}   launches {n}")


if __name__ == "__main__":
    main()
