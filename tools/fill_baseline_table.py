#!/usr/bin/env python
"""BASELINE.md §3 result table from the committed round summaries: python tools/fill_baseline_table.py r03"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
L = lambda n: json.load(open(os.path.join(ROOT, "profiles", "%s_%s" % (tag, n))))
b, v, rc = L("bench.json"), L("bench_vctk.json"), L("bench_one_rank_rccl.json")
d = b["decode"]
rows, cores = b["cpu_baseline"]["rows"], b["cpu_baseline"]["cores"]
table = '''| config | device | batch | mel-frames/s (padded) | mel-frames/s (valid) | %% of roofline | notes |
|---|---|---|---|---|---|---|
| 1. CPU restatement (plumbing) | GPU box host, %d threads (fastest of the 8/16/32/64 sweep; 256 CPUs visible) | 8 | %.0f | %.0f | n/a | PyTorch-CPU fp32 restatement `oracle/torch_ref.py`, FULL train step (fwd + loss + bwd + clip + TF-Adam), 1 warm-up + 5 timed, median %.2f s/step |
| 1b. CPU restatement | same | 32 | %.0f | %.0f | n/a | same shapes as the GPU run, 1 warm-up + 3 timed, median %.2f s/step |
| 2. bf16, 1x MI355X | gfx950 | 32 | **%.0f** | %.0f | whole step %.2f %% of the bf16 MFMA peak (343.6 GFLOP in %.2f ms); dominant kernel `attn_cluster_bwd_k` %.2f %% | `profiles/%s_bench.json`: %.2f ms/step (r2 9.22, r1 10.56) |
| 3. bf16 DP, 2/4/8x MI355X | gfx950, RCCL/xGMI | 32 per GPU | not measured | - | - | one GPU at a time for the builder; the driver's SCALE run was skipped in rounds 1-2.  One-rank RCCL group on one GPU: %.2f ms/step (`profiles/%s_bench_one_rank_rccl.json`) |
| 4. VCTK multi-speaker | 1x MI355X | 32 | %.0f | %.0f | - | config 4's own shape (Ti=80, Tm=500, 152 speakers): %.2f ms/step (`profiles/%s_bench_vctk.json`) |
| 5. free-running inference | 1x MI355X | 1 | %.0f frames/s, RTF %.4f | - | - | hipGraph of %d launches per decoder step, KV cache: %.1f us per step (`decode` of `profiles/%s_bench.json`) |

(Measured by `python bench.py` / `tools/final_measure.sh` on the %s build; the TF1 reference itself cannot be timed, see section 2.)

''' % (cores, rows[0]["padded_mel_frames_per_sec"], rows[0]["valid_mel_frames_per_sec"], rows[0]["s_per_step_median"],
       rows[1]["padded_mel_frames_per_sec"], rows[1]["valid_mel_frames_per_sec"], rows[1]["s_per_step_median"],
       b["value"], b["valid_mel_frames_per_sec"], 100 * b["step_frac_of_bf16_peak"], b["ms_per_step"], 100 * b["roofline"]["frac"], tag,
       b["ms_per_step"], rc["ms_per_step"], tag, v["value"], v["valid_mel_frames_per_sec"], v["ms_per_step"], tag,
       d["mel_frames_per_sec"], d["realtime_factor"], d["launches_per_step"], 1e3 * d["ms_per_step"], tag, tag)
p = os.path.join(ROOT, "BASELINE.md")
s = open(p).read()
i = s.index("| config | device | batch | mel-frames/s (padded)")
j = s.index("## 4. Roofline arithmetic")
open(p, "w").write(s[:i] + table + s[j:])
print(table)
