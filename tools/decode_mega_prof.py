#!/usr/bin/env python
"""Per-phase time of the persistent decode step (csrc/decode_mega2.hip built with -DSATT_MEGA_PROF: tools/build_variant.sh megaprof
decode_mega2.hip -DSATT_MEGA_PROF; SATT_LIB_PATH=tools/probes/libsatt_megaprof.so).  Workgroup 0's wall-clock sums per phase / step."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import satt_amd  # noqa: F401
from satt_amd import ops, _lib
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.inference import infer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
WG = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # the workgroup whose phases are recorded (second form)
ops.set_precision("bf16")
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
g = np.random.default_rng(1234)
src = g.integers(1, 68, (B, 100)); src[:, 0] = 0; src[:, -1] = 0
sl = np.full((B,), 100, dtype=np.int64)
steps = 200
infer(eng, src, sl, max_steps=steps, min_steps=10 ** 6)
l = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
read = l.satt_dec_mega2_prof_read
l.satt_dec_mega2_prof_select(WG)
read(buf, 1)
out = infer(eng, src, sl, max_steps=steps, min_steps=10 ** 6)
torch.cuda.synchronize()
read(buf, 0)
us = [x / 100.0 / steps for x in buf]
print("B=%d: %.2f us per step (HIP events); workgroup %d phases, us per step:" % (B, out["decode_ms"] * 1e3 / steps, WG))
if True:
    names = ["loop top (+ teacher frame)", "pre-net 0 (split)", "x p0", "pre-net 1 (split)", "x p1", "attention LSTM slice + cell", "x hq",
             "query layer (split)", "x pq", "energies", "x e1|e2 + softmax + recursion", "LSTM 1: context tables + slice + cell",
             "x h1", "LSTM 2 slice + cell", "x dout", "K|V|Q slice", "self-attention partial (x kvq row)", "x partials", "merge",
             "output transform (split)", "x tr", "projection (split)", "x y", "step tail"]
    for n, v in zip(names, us):
        print("  %-36s %6.2f" % (n, v))
    for k, n in ((24, "e: tables requested, energies polled"), (25, "e: softmax + recursion"), (26, "F: rows requested, q polled"), (27, "F: barrier"),
                 (28, "F: scores"), (29, "F: statistics + P V")):
        print("    (sub-mark, included above) %-36s %6.2f" % (n, us[k]))
    print("    (wave 7 of workgroup 0: its energy row, request to publication) %6.2f" % us[31])
    print("  shader clock over the steps: %.0f MHz (s_memtime ticks / wall clock)" % (buf[30] / max(sum(buf[:30]), 1) * 100.0))
    print("  sum %.2f, of which exchanges (x ...) %.2f" % (sum(us[:30]), sum(v for n, v in zip(names, us) if n.startswith("x "))))
