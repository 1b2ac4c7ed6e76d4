#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc result databases (one per pass) into per-kernel counter averages."""
import glob
import sqlite3
import sys


def one(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    t = lambda like: [r[0] for r in cur.execute("select name from sqlite_master where name like '%s%%'" % like)][0]
    sym, dsp, info, ev = t("rocpd_info_kernel_symbol"), t("rocpd_kernel_dispatch"), t("rocpd_info_pmc"), t("rocpd_pmc_event")
    cols = [c[1] for c in cur.execute(f"pragma table_info({ev})")]
    icol = [c[1] for c in cur.execute(f"pragma table_info({info})")]
    q = (f"select s.kernel_name, i.name, count(*), avg(e.value), sum(e.value) from {ev} e "
         f"join {info} i on e.pmc_id=i.id join {dsp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id "
         f"group by s.kernel_name, i.name")
    try:
        return list(cur.execute(q))
    except Exception as ex:
        print("query failed", ex, cols, icol)
        return []


def main(pattern):
    rows = []
    import os
    paths = sorted(set(glob.glob(pattern)) | set(glob.glob(pattern.replace('/*.db', '/*/*.db'))))
    for p in paths:
        rows += one(p)
    ker = {}
    for k, c, n, avg, tot in rows:
        ker.setdefault(k.replace(".kd", ""), {})[c] = (n, avg)
    for k in sorted(ker):
        if not k.startswith("tbz_"):
            continue
        print(k)
        for c in sorted(ker[k]):
            n, avg = ker[k][c]
            print("    %-28s launches=%d avg_per_launch=%.4g" % (c, n, avg))


if __name__ == "__main__":
    main(sys.argv[1])
