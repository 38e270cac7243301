#!/usr/bin/env python
"""Run-to-run spread of the parameter gradients: N fresh engines, same parameters / batch / seed, one train step each;
per tensor: max |g_i - g_0| relative to max |g_0| (atomics reorder sums; anything above ~1e-5 is a missing dependency)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import satt_amd  # noqa: F401
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig, init_params
from satt_amd.datasets.synthetic import synthetic_batch

ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
cfg = ModelConfig()
P = init_params(cfg, 1)
batch = synthetic_batch(32, 160, 800, num_mels=cfg.num_mels, r=cfg.r, seed=5)
ref = None
for i in range(4):
    eng = Engine(cfg, "cuda", params=P, rng_seed=3)
    ctx = eng.train_step(eng.to_device_batch(batch))
    torch.cuda.synchronize()
    eng.check_clusters(ctx)
    g = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in eng.G.items()}
    if ref is None:
        ref = g
        continue
    rows = sorted(((float(np.abs(g[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)), k) for k in g), reverse=True)
    print("run %d vs run 0: worst tensors:" % i, ", ".join("%s %.2e" % (k, e) for e, k in rows[:6]))
