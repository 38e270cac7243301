#!/usr/bin/env python
"""Time-ordered kernel dispatches (> min_us) of the LAST training step of a rocprofv3 run (rocpd sqlite)."""
import glob, os, sqlite3, sys
src = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[-1]
db = sqlite3.connect(src); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
qcol = "d.queue_id" if "queue_id" in cols else "0"
rows = cur.execute("select s.kernel_name, d.end-d.start, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.start, %s "
                   "from %s d join %s s on d.kernel_id=s.id order by d.start" % (qcol, kd, ks)).fetchall()
idx = [i for i, r in enumerate(rows) if "adam_k" in r[0]]
lo, hi = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
t0 = rows[lo][6]
print("dispatches in last step:", hi - lo, " wall: %.3f ms" % ((rows[hi - 1][6] + rows[hi - 1][1] - t0) / 1e6))
small = 0.0; nsmall = 0
for r in rows[lo:hi]:
    if r[1] / 1e3 < min_us:
        small += r[1] / 1e3; nsmall += 1
        continue
    n = r[0].replace("_ZN12_GLOBAL__N_1", "").replace(".kd", "")[:46]
    print("%8.3f ms  %-46s %8.1f us  grid=(%d,%d,%d) q=%s" % ((r[6] - t0) / 1e6, n, r[1] / 1e3, r[2] // r[5], r[3], r[4], r[7]))
print("(%d dispatches below %.0f us, total %.1f us)" % (nsmall, min_us, small))
