import sys
sys.path.insert(0, '.')
import numpy as np, torch
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
B = int(sys.argv[1]); prec = sys.argv[2]
batch = synthetic_batch(B, 160, 800, seed=77)
ops.set_precision(prec)
def fwd(eng, b):
    eng.zero_grad(); ctx = eng.forward(b, True); torch.cuda.synchronize()
    return ctx, {k: ctx[k].detach().clone() for k in ("att_out", "h1", "dec_out")}
e0 = Engine(ModelConfig(), "cuda", param_seed=3, rng_seed=5); e0.single_launch_attention = False
b = e0.to_device_batch(batch)
_, ref = fwd(e0, b)
e1 = Engine(ModelConfig(), "cuda", param_seed=3, rng_seed=5); e1.single_launch_attention = True
calls = []
orig = ops.stream_wait_value
def spy(counter, value, stream=None):
    calls.append(value); return orig(counter, value, stream)
ops.stream_wait_value = spy
ctx, t1 = fwd(e1, b)
print("B", B, prec, "wait values", calls, "counter after fwd", int(e1._keep_fwd[0]), "bounds", e1._chunk_bounds(400, e1.pipeline_chunks))
d = (t1["h1"] - ref["h1"]).abs().view(B, 400, -1).amax(-1)
print("h1 max err", float(d.max()), "first bad step per sample:", [int((d[i] > 1e-3).nonzero()[0]) if bool((d[i] > 1e-3).any()) else -1 for i in range(B)])
