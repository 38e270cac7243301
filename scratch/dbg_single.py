import sys
sys.path.insert(0, '.')
import numpy as np, torch
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
batch = synthetic_batch(4, 160, 800, seed=77)
def run(prec, single):
    ops.set_precision(prec)
    eng = Engine(ModelConfig(), "cuda", param_seed=3, rng_seed=5)
    eng.single_launch_attention = single
    b = eng.to_device_batch(batch)
    eng.zero_grad(); ctx = eng.forward(b, True); eng.backward(ctx); torch.cuda.synchronize()
    try:
        eng.check_clusters(ctx)
    except Exception as e:
        print("cluster status:", e)
    return {k: v.detach().cpu().numpy().astype(np.float64).ravel() for k, v in eng.G.items()}, float(eng.losses[2])
def cos(a, b): return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
ref = {}
import os
if os.environ.get('WARM'):
    w = torch.ones(4, dtype=torch.int32, device='cuda'); st = Engine._device_streams(torch.device('cuda'))[0]
    ops.stream_wait_value(w[0:1], 1, st); ops.stream_write_value(w[1:2], 5, st); torch.cuda.synchronize(); print('warm', w.tolist())
for prec in ("f32", "bf16"):
    g0, l0 = run(prec, False)
    for rep in range(2):
        g1, l1 = run(prec, True)
        bad = {k: round(cos(g0[k], g1[k]), 4) for k in g0 if cos(g0[k], g1[k]) < 0.999}
        print(prec, "loss", l0, l1, "tensors with cos<0.999:", len(bad), list(bad.items())[:8])
