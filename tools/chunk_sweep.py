#!/usr/bin/env python
"""Sweep of the layer-pipeline chunking (Engine.pipeline_chunks x pipeline_tail) on the benchmark step: ms per step, median of
`reps` timed steps after warm-up, one engine per setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import satt_amd  # noqa: F401
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch

ops.set_precision("bf16")
cfg = ModelConfig()
batch = synthetic_batch(32, 160, 800, num_mels=cfg.num_mels, r=cfg.r, seed=5)
eng = Engine(cfg, "cuda", rng_seed=3)
b = eng.to_device_batch(batch)


def run(nc, tail, reps=30):
    eng.pipeline_chunks, eng.pipeline_tail = nc, tail
    for _ in range(6):
        eng.train_step(b); eng.optimizer_step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        eng.train_step(b); eng.optimizer_step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


# (per-step host sync: the absolute numbers are ~0.1 ms above the free-running bench; the ranking is what matters)
for nc in (5, 6, 7, 8):
    for tail in ((3, 4), (4, 4), (3, 6), (4, 6), (2, 4), (3, 3)):
        print("chunks %d tail %s: %.3f ms" % (nc, tail, run(nc, tail)), flush=True)
