#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gp; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for d in 0; do
  SATT_GEMM_DEEP_MAX=$d timeout 200 rocprofv3 --kernel-trace -d $O/t$d -- python $R/scratch/gemm_rows_probe.py > $O/log$d.txt 2>&1
  python - <<PY
import glob, sqlite3
db = sqlite3.connect(sorted(glob.glob("$O/t$d/**/*.db", recursive=True))[-1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute("select s.kernel_name, d.grid_size_x/d.workgroup_size_x, d.grid_size_y, d.grid_size_z, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from %s d join %s s on d.kernel_id=s.id where s.kernel_name like '%%gemm%%' group by 1,2,3,4 order by 2,3,4" % (kd, ks)).fetchall()
print("DEEP_MAX=$d")
for r in rows: print("  %-40s grid=(%d,%d,%d) n=%d avg %.1f us min %.1f us" % (r[0][21:58], r[1], r[2], r[3], r[4], r[5], r[6]))
PY
done
