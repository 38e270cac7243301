#!/usr/bin/env python
"""List the longest individual kernel dispatches of a rocprofv3 run (rocpd sqlite), excluding name patterns."""
import glob, os, sqlite3, sys
src = sys.argv[1]
excl = sys.argv[2].split(",") if len(sys.argv) > 2 else []
if os.path.isdir(src):
    src = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)[0]
db = sqlite3.connect(src); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute("select s.kernel_name, d.end-d.start, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.start "
                   "from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
# keep the last step only: find last adam_k and the one before
idx = [i for i, r in enumerate(rows) if "adam_k" in r[0]]
lo, hi = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
step = [r for r in rows[lo:hi] if not any(e in r[0] for e in excl)]
t0 = rows[lo][6]
print("dispatches in last step:", hi - lo, " wall (first start -> last end): %.3f ms" % ((rows[hi - 1][6] + rows[hi - 1][1] - t0) / 1e6))
for r in sorted(step, key=lambda r: -r[1])[:40]:
    n = r[0].replace("_ZN12_GLOBAL__N_1", "").replace(".kd", "")[:60]
    print("%-62s %9.1f us  grid=(%d,%d,%d)/%d  @%.2f ms" % (n, r[1] / 1e3, r[2] // r[5], r[3], r[4], r[5], (r[6] - t0) / 1e6))
