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
def run(**kw):
    eng = Engine(ModelConfig(), "cuda", param_seed=5, rng_seed=9)
    for k, v in kw.items(): setattr(eng, k, v)
    b = eng.to_device_batch(batch)
    for _ in range(2):
        eng.zero_grad(); ctx = eng.forward(b, True); eng.backward(ctx)
    torch.cuda.synchronize(); eng.check_clusters(ctx)
    return dict(U=eng.G["dec.att1.U"].double().cpu().numpy().copy(), fl=ctx["flb"].double().cpu().numpy().copy(),
                saf=ctx["saf"].float().cpu().numpy().copy() if "saf" in ctx else None, b=eng.G["dec.att1.b"].double().cpu().numpy().copy())
def d(x, y): return float(np.abs(x - y).max() / (np.abs(y).max() + 1e-30))
for kw in (dict(), dict(single_launch_attention=False), dict(overlap_wgrad=False), dict(pg_lds_pad=0)):
    r = [run(**kw) for _ in range(3)]
    lens = batch["source_length"]
    print(kw, "dU", ["%.1e" % d(r[i]["U"], r[0]["U"]) for i in (1, 2)], "db", ["%.1e" % d(r[i]["b"], r[0]["b"]) for i in (1, 2)],
          "fl", ["%.1e" % d(r[i]["fl"], r[0]["fl"]) for i in (1, 2)],
          "saf(valid rows)", ["%.1e" % max(d(r[i]["saf"].reshape(B, -1, Ti, 256)[bb, :, :int(lens[bb])], r[0]["saf"].reshape(B, -1, Ti, 256)[bb, :, :int(lens[bb])]) for bb in range(B)) for i in (1, 2)])
