#!/usr/bin/env python
"""Per-kernel totals of one rocprofv3 --pmc counter (rocpd sqlite): python tools/rocprof_pmc.py <dir-or-db> <COUNTER> [top]
Prints, per kernel name: dispatches, sum and mean of the counter value per dispatch (raw units of the counter)."""
import glob, os, sqlite3, sys
src, cname = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[-1]
db = sqlite3.connect(src); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
kcols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
ids = [r[0] for r in cur.execute("select id from %s where name=? or symbol=?" % pi, (cname, cname))]
units = cur.execute("select units, description from %s where name=? limit 1" % pi, (cname,)).fetchone()
print("counter %s units=%r (%s)" % (cname, units[0] if units else None, (units[1] if units else "")[:80]))
q = ("select s.kernel_name, count(distinct d.id), sum(e.value) from %s e join %s d on e.event_id=d.event_id "
     "join %s s on d.kernel_id=s.id where e.pmc_id in (%s) group by s.kernel_name order by sum(e.value) desc"
     % (pe, kd, ks, ",".join(str(i) for i in ids))) if "event_id" in kcols else None
rows = cur.execute(q).fetchall()
tot = sum(r[2] for r in rows)
print("total over all kernels: %.1f" % tot)
for n, cnt, v in rows[:top]:
    n = n.replace("_ZN12_GLOBAL__N_1", "").replace(".kd", "")[:58]
    print("%-60s n=%5d  sum=%14.1f  per-dispatch=%12.2f" % (n, cnt, v, v / max(cnt, 1)))
