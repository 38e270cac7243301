"""which elements of the flat gradient buffer differ between two identical runs (same seeds, same batch)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
ops.set_precision("bf16")
B, Ti, Tm = 3, 160, 200
batch = synthetic_batch(B, Ti, Tm, seed=77)
def run():
    eng = Engine(ModelConfig(), "cuda", param_seed=5, rng_seed=9)
    for kv in filter(None, os.environ.get("SATT_SET", "").split(";")):
        k, v = kv.split("="); setattr(eng, k, eval(v))
    b = eng.to_device_batch(batch)
    eng.zero_grad(); ctx = eng.forward(b, True); eng.backward(ctx)
    torch.cuda.synchronize(); eng.check_clusters(ctx)
    return eng, eng.grad.detach().double().cpu().numpy().copy()
eng, g0 = run()
names = sorted(eng.layout.items(), key=lambda kv: kv[1][0])
def where(i):
    for k, (o, shp) in names:
        n = int(np.prod(shp))
        if o <= i < o + n: return "%s[%d]" % (k, i - o)
    return "?"
for rep in range(4):
    _, g1 = run()
    d = np.abs(g1 - g0)
    # noise floor per tensor: 1e-5 of the tensor's max
    bad = []
    for k, (o, shp) in names:
        n = int(np.prod(shp)); seg = d[o:o + n]; m = np.abs(g0[o:o + n]).max()
        idx = np.nonzero(seg > 1e-4 * m + 1e-30)[0]
        if len(idx): bad.append((k, len(idx), int(idx.min()), int(idx.max()), float(seg.max() / (m + 1e-30)), m))
    print("rep", rep, [(k, n, lo, hi, "%.1e" % r, "max %.1e" % m) for k, n, lo, hi, r, m in bad][:8])
