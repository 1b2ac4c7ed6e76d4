#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats results .db (rocpd sqlite) into the per-kernel table
kept under profiles/ (avg/min/max duration per kernel, launches, grid, LDS, registers)."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    sym = [r[0] for r in cur.execute("select name from sqlite_master where name like 'rocpd_info_kernel_symbol%'")][0]
    dsp = [r[0] for r in cur.execute("select name from sqlite_master where name like 'rocpd_kernel_dispatch%'")][0]
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         f"sum(d.end-d.start), max(d.grid_size_x), max(d.workgroup_size_x), max(s.group_segment_size), "
         f"max(s.arch_vgpr_count), max(s.sgpr_count), max(s.accum_vgpr_count), max(s.private_segment_size) "
         f"from {dsp} d join {sym} s on d.kernel_id=s.id "
         f"group by s.kernel_name order by 6 desc")
    rows = list(cur.execute(q))
    tot = sum(r[5] for r in rows) or 1
    # vgpr / agpr: arch and accumulation VGPRs as the code object's metadata has them (the two share one file of 512 per
    # SIMD lane: their sum, rounded up to 8, is what bounds the waves per SIMD); scr = scratch octets per lane (spills)
    lines = ["%-34s %6s %12s %12s %12s %7s %10s %5s %7s %5s %5s %5s %5s" %
             ("kernel", "calls", "avg_us", "min_us", "max_us", "pct", "grid", "wg", "lds_B", "vgpr", "agpr", "sgpr", "scr")]
    for r in rows:
        name = r[0].replace(".kd", "")
        if len(name) > 34:
            name = name[:31] + "..."
        lines.append("%-34s %6d %12.1f %12.1f %12.1f %6.1f%% %10d %5d %7d %5d %5d %5d %5d" %
                     (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / tot, r[6], r[7], r[8], r[9], r[11] or 0, r[10],
                      r[12] or 0))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
