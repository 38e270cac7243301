#!/usr/bin/env python
"""Launch-by-launch timeline of the decode step from a rocprofv3 kernel trace of tools/bench_infer.py:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $REPO/tools/bench_infer.py --steps 64
    python tools/decode_timeline.py $OUT            # prints two consecutive steps of the LAST graph replay
Times in us relative to the first launch shown; `gap` = start minus the previous launch's end."""
import csv, glob, os, sys

src = sys.argv[1]
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = [r for r in csv.DictReader(open(src))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


dec = [r for r in rows if "dec_" in r["Kernel_Name"]]
if not dec:
    sys.exit("no decode kernels in the trace")
# a step starts at the launch that follows the attention-context kernel's predecessor chain: find the period by the
# positions of the energy kernel
idx = [i for i, r in enumerate(dec) if "dec_attn_energy_k" in r["Kernel_Name"]]
if len(idx) < 4:
    sys.exit("fewer than 4 decoder steps in the trace")
lo, hi = idx[-3] + 1, idx[-1] + 1          # two whole steps ending with the last energy launch
t0 = int(dec[lo]["Start_Timestamp"])
prev = None
print("%-46s %9s %9s %7s %6s  grid" % ("launch", "start", "end", "dur", "gap"))
for r in dec[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev is None else (s - prev) / 1e3
    prev = e
    g = "%dx%d" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
    print("%-46s %9.2f %9.2f %7.2f %6.2f  %s x %s" % (short(r["Kernel_Name"]), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap, g,
                                                     r["Workgroup_Size_X"]))
n = (hi - lo) // 2
print("launches per step: %d   step period: %.2f us" % (n, (int(dec[hi - 1]["End_Timestamp"]) - int(dec[lo - 1]["End_Timestamp"])) / 2e3))
