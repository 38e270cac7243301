#!/usr/bin/env python
"""Repeat the free-running decode of the frozen b1 / b2 fixtures N times in one process and count runs whose mel differs from the
first run (the persistent step kernel is deterministic: every run must be bit-identical).  python tools/decode_stress.py [N] [case]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import satt_amd  # noqa
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig, init_params
from satt_amd.inference import infer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
case = sys.argv[2] if len(sys.argv) > 2 else "b1"
fresh = len(sys.argv) > 3 and sys.argv[3] == "fresh"       # a new engine + session per run (every utterance is a session's first)
z = np.load(os.path.join(ROOT, "tests", "golden", "decode_ljspeech_%s.npz" % case))
cfg = ModelConfig(); P = dict(init_params(cfg, int(z["param_seed"])))
ops.set_precision("bf16")


def make():
    eng = Engine(cfg, "cuda", params=P, rng_seed=7)
    for name, (mean, var) in eng.bn.items():
        mean.copy_(torch.as_tensor(z["bn_mean." + name])); var.copy_(torch.as_tensor(z["bn_var." + name]))
    return eng


eng = make()
steps = int(z["steps"])
ref, bad, worst = None, 0, 0.0
import time
_t0, _ms = time.time(), []
for i in range(N):
    if fresh and i:
        eng = make()
    out = infer(eng, z["source"], z["source_length"], max_steps=steps, min_steps=10 ** 6, use_graph=True)
    mel = out["mel"].float().cpu().numpy()
    _ms.append(out["decode_ms"])
    if ref is None:
        ref = mel
        print("run 0: |mel - golden| max %.3e" % (np.abs(mel.astype(np.float64) - z["mel"]).max() if "mel" in z.files else float("nan")))
    else:
        d = float(np.abs(mel - ref).max())
        if d != 0.0:
            bad += 1; worst = max(worst, d)
            t = np.abs(mel - ref).reshape(steps, -1).max(-1)
            print("run %d differs: max %.3e, first differing step %d" % (i, d, int(np.argmax(t > 0))))
print("%s: %d runs, %d differ from run 0 (worst %.3e)%s; decode %.1f ms per utterance (median), %.0f s in all" % (case, N, bad, worst, " [fresh session per run]" if fresh else "", sorted(_ms)[len(_ms) // 2], time.time() - _t0))
