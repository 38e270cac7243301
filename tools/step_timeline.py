"""kernel timeline of ONE train step from a rocprofv3 --kernel-trace run (rocpd database): every dispatch between the end of the
previous step's adam_k and this step's adam_k, in start order, with queue, duration and the gap to the previous end on that queue.
python tools/step_timeline.py <dir> [step index, default 3] [--outside]   (--outside: only dispatches that start outside the two
attention cluster kernels, i.e. what the step adds around the loops)"""
import glob, sqlite3, sys
db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[-1]
step = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
outside = "--outside" in sys.argv
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
    print(tabs); sys.exit(1)
cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
rows = list(c.execute("select name, start, end, queue_id, grid_x*grid_y*grid_z, workgroup_x*workgroup_y*workgroup_z from %s order by start" % view)) \
    if "grid_x" in cols else list(c.execute("select name, start, end, queue_id, 0, 0 from %s order by start" % view))
adam = [i for i, r in enumerate(rows) if "adam_k" in r[0]]
lo, hi = adam[step - 1] + 1, adam[step] + 1
seg = rows[lo:hi]
t0 = seg[0][1]
fw = [r for r in seg if "attn_cluster_fwd_k" in r[0]]; bw = [r for r in seg if "attn_cluster_bwd_k" in r[0]]
inside = lambda s: any(a[1] <= s < a[2] for a in fw + bw)
last = {}
for name, s, e, q, g, w in seg:
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")[:58]
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    if outside and inside(s) and "attn_cluster" not in name:
        continue
    print("%9.1f %9.1f  q%-3s dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, (e - s) / 1e3, gap, n))
print("step: %.1f us" % ((seg[-1][2] - t0) / 1e3))
