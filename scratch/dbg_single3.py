import sys
sys.path.insert(0, '.')
import numpy as np, torch
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
batch = synthetic_batch(4, 160, 800, seed=77)
ops.set_precision("f32")
def fwd(eng, b):
    eng.zero_grad(); ctx = eng.forward(b, True); torch.cuda.synchronize()
    keys = ("lstm_out", "sa_out", "att_out", "h1", "dec_out", "yout", "al1")
    return ctx, {k: ctx[k].detach().clone() for k in keys}
e0 = Engine(ModelConfig(), "cuda", param_seed=3, rng_seed=5); e0.single_launch_attention = False
b = e0.to_device_batch(batch)
_, ref = fwd(e0, b)
e1 = Engine(ModelConfig(), "cuda", param_seed=3, rng_seed=5); e1.single_launch_attention = True
ctx, t1 = fwd(e1, b)
print("fwd#1 single vs multi:", {k: float((t1[k] - ref[k]).abs().max()) for k in ref})
e1.backward(ctx); torch.cuda.synchronize()
try:
    e1.check_clusters(ctx); print("clusters ok")
except Exception as ex:
    print("cluster status", ex)
ctx, t2 = fwd(e1, b)
print("fwd#2 single vs multi:", {k: float((t2[k] - ref[k]).abs().max()) for k in ref})
d = (t2["att_out"] - ref["att_out"]).abs().view(4, 400, -1).amax(-1)
print("first bad step per sample:", [int((d[i] > 1e-4).nonzero()[0]) if bool((d[i] > 1e-4).any()) else -1 for i in range(4)])
_, t3 = fwd(e0, b)
print("multi engine again vs ref:", {k: float((t3[k] - ref[k]).abs().max()) for k in ref})
