#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db or *_kernel_stats.csv) as a text table:
per-kernel call count, total / average duration and share.   usage: rocprof_summary.py <dir-or-db> [steps]"""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
         "max(d.end-d.start), max(s.arch_vgpr_count), max(d.group_segment_size), max(d.workgroup_size_x), "
         "max(d.grid_size_x) from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kd, ks))
    return cur.execute(q).fetchall()


def main():
    src = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if os.path.isdir(src):
        dbs = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
        src = dbs[0]
    rows = from_db(src)
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace summary of %s" % os.path.basename(src))
    print("# total kernel time %.3f ms over %d kernel names%s" %
          (tot / 1e6, len(rows), (" (%.3f ms per train step, %d steps)" % (tot / 1e6 / steps, steps)) if steps else ""))
    print("%-72s %6s %11s %11s %10s %10s %6s %5s %7s %8s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us",
                                                         "share", "vgpr", "lds_B", "grid_x"))
    for n, c, s, a, mn, mx, vg, lds, wg, gx in rows:
        n = n.replace("_ZN12_GLOBAL__N_1", "").replace(".kd", "")
        print("%-72s %6d %11.3f %11.1f %10.1f %10.1f %5.1f%% %5s %7s %8s" %
              (n[:72], c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot, vg, lds, gx))


if __name__ == "__main__":
    main()
