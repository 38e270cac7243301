"""per-kernel totals of a rocprofv3 --kernel-trace --stats run (rocpd database): python tools/kstats.py <dir> [steps] [filter ...]"""
import glob, sqlite3, sys
db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
flt = sys.argv[3:]
c = sqlite3.connect(db)
for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
    n = str(name).replace("(anonymous namespace)::", "").replace("void ", "")
    if flt and not any(f in n for f in flt):
        continue
    print("%-64s %6d calls %10.1f us  avg %8.2f  per step %8.1f" % (n[:64], calls, total, avg, total / steps))
