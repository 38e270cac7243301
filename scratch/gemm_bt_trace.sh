#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gbt; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $O/t -- python $R/scratch/gemm_bt_probe.py > $O/log.txt 2>&1
grep "max diff" $O/log.txt
python - <<PY
import glob, sqlite3
db = sqlite3.connect(sorted(glob.glob("$O/t/**/*.db", recursive=True))[-1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute("select s.kernel_name, d.grid_size_x/d.workgroup_size_x, d.grid_size_y, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from %s d join %s s on d.kernel_id=s.id where s.kernel_name like '%%gemm%%' group by 1,2,3 order by 2,3,1" % (kd, ks)).fetchall()
for r in rows: print("  %-34s grid=(%d,%d) n=%d avg %.1f us min %.1f us" % (r[0][21:55], r[1], r[2], r[3], r[4], r[5]))
PY
