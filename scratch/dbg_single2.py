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
order = [bool(int(x)) for x in sys.argv[1]]
for single in order:
    eng = Engine(ModelConfig(), "cuda", param_seed=3, rng_seed=5)
    eng.single_launch_attention = single
    b = eng.to_device_batch(batch)
    for rep in range(2):
        eng.zero_grad(); ctx = eng.forward(b, True); torch.cuda.synchronize()
        l = float(eng.losses[2])
        eng.backward(ctx); torch.cuda.synchronize()
        g = eng.grad.double()
        print("single" if single else "multi ", "loss %.7f" % l, "gradnorm %.6f" % float(g.norm()), "gsum %.6f" % float(g.sum()))
