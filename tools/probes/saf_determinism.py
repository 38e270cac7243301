"""run-to-run differences of the gradients with / without the saved attention factors (same seeds, same batch)"""
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
def run(on):
    eng = Engine(ModelConfig(), "cuda", param_seed=5, rng_seed=9)
    eng.save_attention_factors = on
    b = eng.to_device_batch(batch)
    for _ in range(2):
        eng.zero_grad(); ctx = eng.forward(b, True); eng.backward(ctx)
    torch.cuda.synchronize(); eng.check_clusters(ctx)
    return {k: v.detach().double().cpu().numpy() for k, v in eng.G.items()}
def cmp(a, b, tag):
    rows = sorted(((float(np.abs(a[k] - b[k]).max() / (np.abs(b[k]).max() + 1e-30)), k) for k in a), reverse=True)[:4]
    print(tag, ["%s %.2e" % (k, e) for e, k in rows])
for rep in range(3):
    a1, a2 = run(True), run(True)
    b1, b2 = run(False), run(False)
    cmp(a1, a2, "saf vs saf  :")
    cmp(b1, b2, "base vs base:")
    cmp(a1, b1, "saf vs base :")
