"""unsynchronised train steps over batches whose (B, Ti, Tm) changes from step to step (what a length-bucketed corpus feeds):
hand-off timeouts, finite losses, ms per step against the sum of the per-shape steady-state times"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import satt_amd  # noqa: F401
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
ops.set_precision("bf16")
g = np.random.default_rng(int(os.environ.get("SEED", "1")))
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
shapes = []
for _ in range(int(os.environ.get("NSHAPES", "12"))):
    Tm = int(g.integers(60, 401)) * 2
    Ti = int(np.clip(Tm // 5 + g.integers(-10, 11), 12, 160))
    B = int(g.choice([32, 32, 32, 16, 24, 8]))
    shapes.append((B, Ti, Tm))
batches = [eng.to_device_batch(synthetic_batch(B, Ti, Tm, seed=100 + i)) for i, (B, Ti, Tm) in enumerate(shapes)]
for b in batches:                       # first visit of every shape (allocations, packs)
    ctx = eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize(); eng.check_clusters(ctx)
per = []
for b in batches:                       # steady state per shape
    for _ in range(2): eng.train_step(b); eng.optimizer_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): ctx = eng.train_step(b); eng.optimizer_step()
    torch.cuda.synchronize(); per.append((time.perf_counter() - t0) / 4 * 1e3)
eng.check_clusters(ctx)
order = [int(i) for i in g.integers(0, len(batches), 80)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in order:
    ctx = eng.train_step(batches[i]); eng.optimizer_step()
torch.cuda.synchronize(); mixed = (time.perf_counter() - t0) * 1e3
eng.check_clusters(ctx)
expect = sum(per[i] for i in order)
for s, p in zip(shapes, per): print("B=%d Ti=%d Tm=%d: %.2f ms" % (*s, p))
print("80 steps in random shape order: %.1f ms, sum of the per-shape steady-state times %.1f ms (ratio %.3f), loss %.4f"
      % (mixed, expect, mixed / expect, float(eng.losses[2])))
