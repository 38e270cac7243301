#!/usr/bin/env python
"""Launch-by-launch timeline of ONE train step from a rocprofv3 kernel trace of bench.py:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $REPO/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-decode
    python tools/step_timeline.py $OUT [phase]        # phase: all | enc_fwd | head | enc_bwd (default all)
The step shown is the last complete one (from one embedding_fwd_k to the next).  Per launch: hardware queue, start offset, duration,
gap to the previous launch of the SAME queue, and `idle` = time during which NO kernel of any queue was running before this launch
started (a gap of the whole device: launch latency or a dependency on the host).  The summary gives the device-idle total per phase."""
import csv, glob, os, sys

src = sys.argv[1]
phase = sys.argv[2] if len(sys.argv) > 2 else "all"
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = [r for r in csv.DictReader(open(src))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:52]


starts = [i for i, r in enumerate(rows) if "embedding_fwd_k" in r["Kernel_Name"]]
if len(starts) < 2:
    sys.exit("fewer than two steps in the trace")
lo, hi = starts[-2], starts[-1]
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
qids = sorted({r["Queue_Id"] for r in step})
qname = {q: "q%d" % i for i, q in enumerate(qids)}
# phase boundaries by marker kernels
def first(name, after=0):
    for i, r in enumerate(step):
        if i >= after and name in r["Kernel_Name"]:
            return i
    return None
i_attn_f = first("attn_cluster_fwd_k")
i_loss = first("loss_fused_k") or first("loss")
i_attn_b = first("attn_cluster_bwd_k")
bounds = {"enc_fwd": (0, i_attn_f), "head": (i_loss, i_attn_b), "all": (0, len(step))}
# encoder backward: from the end of the backward attention kernel to the end of the step
if i_attn_b is not None:
    tb = int(step[i_attn_b]["End_Timestamp"])
    k = next((i for i, r in enumerate(step) if int(r["Start_Timestamp"]) >= tb), len(step))
    bounds["enc_bwd"] = (k, len(step))
a, b = bounds.get(phase, bounds["all"])
prev_end = {}
busy_until = 0
idle_total = 0.0
print("%-54s %4s %9s %8s %7s %7s  grid" % ("launch", "q", "start_us", "dur_us", "qgap", "idle"))
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r["Queue_Id"]
    qgap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    idle = max(0.0, (s - busy_until) / 1e3) if busy_until else 0.0
    if a <= i < b:
        idle_total += idle
        g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))) * \
            max(1, int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
        print("%-54s %4s %9.1f %8.1f %7.1f %7.1f  %d" % (short(r["Kernel_Name"]), qname[q], (s - t0) / 1e3, (e - s) / 1e3, qgap, idle, g))
    prev_end[q] = e
    busy_until = max(busy_until, e)
span = (int(step[b - 1]["End_Timestamp"]) - int(step[a]["Start_Timestamp"])) / 1e3
print("phase %s: %d launches, span %.1f us, device idle inside it %.1f us; step period %.1f us" %
      (phase, b - a, span, idle_total, (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3))
