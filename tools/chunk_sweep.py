#!/usr/bin/env python
"""Sweep of the layer-pipeline chunking (Engine.pipeline_chunks x pipeline_tail x pipeline_growth x pipeline_tail_fwd) on the
benchmark step: ms per step, median of `reps` timed steps after warm-up, one engine for all settings (per-step host sync: the
absolute numbers are ~0.1 ms above the free-running bench; the ranking is what matters)."""
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


def run(nc, tail, growth, tail_fwd, reps=24):
    eng.pipeline_chunks, eng.pipeline_tail, eng.pipeline_growth, eng.pipeline_tail_fwd = nc, tail, growth, tail_fwd
    for _ in range(5):
        eng.train_step(b); eng.optimizer_step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx = eng.train_step(b); eng.optimizer_step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    eng.check_clusters(ctx)
    return float(np.median(ts))


if "--growth" in sys.argv:
    for rep in range(2):
        for g in (1.4, 1.6, 1.8, 2.0, 2.4):
            for tail in ((6, 3), (5, 3), (4, 3)):
                print("chunks 8 tail %s growth %.1f: %.3f ms" % (tail, g, run(8, tail, g, None)), flush=True)
    sys.exit(0)
if "--streams" in sys.argv:          # the two decoder LSTM layers on one stream / on two (Engine.lstm_one_stream)
    for rep in range(3):
        for one in (True, False):
            eng.lstm_one_stream = one
            for tail in ((6, 3), (4, 3)):
                for g in (1.3, 1.6, 2.0, 2.5):
                    print("chunks 8 tail %s growth %.1f one_stream %s: %.3f ms" % (tail, g, one, run(8, tail, g, None, reps=12)), flush=True)
    sys.exit(0)
if "--two-bwd" in sys.argv:          # two LSTM streams: backward tails (smallest chunk 8 / 12 / 16 steps), forward tail fixed
    for rep in range(3):
        for one in (True, False):
            eng.lstm_one_stream = one
            for tail in ((6, 3), (7, 4), (7, 6), (8, 6)):
                for g in (1.4, 1.6):
                    print("tail %s growth %.1f one_stream %s: %.3f ms" % (tail, g, one, run(8, tail, g, (6, 3), reps=12)), flush=True)
    sys.exit(0)
if "--two" in sys.argv:              # two LSTM streams: forward tails / growth
    for rep in range(3):
        for one in (True, False):
            eng.lstm_one_stream = one
            for tf in (None, (6, 4), (7, 6), (5, 6)):
                for g in (1.4, 1.6):
                    print("tail_fwd %s growth %.1f one_stream %s: %.3f ms" % (tf, g, one, run(8, (6, 3), g, tf, reps=12)), flush=True)
    sys.exit(0)
if "--wide" in sys.argv:
    for rep in range(2):
        for nc in (6, 8, 10):
            for tail in ((6, 3), (5, 3), (4, 3), (6, 4), (3, 3)):
                for g in (1.6, 2.0, 2.5, 3.0):
                    print("chunks %d tail %s growth %.1f: %.3f ms" % (nc, tail, g, run(nc, tail, g, None, reps=12)), flush=True)
    sys.exit(0)
if "--tdiv" in sys.argv:
    for rep in range(3):
        for nc in (8, 10):
            for tail in ((6, 3), (7, 4), (8, 6), (6, 6), (5, 4)):
                for g in (1.6, 2.0):
                    print("chunks %d tail %s growth %.1f: %.3f ms" % (nc, tail, g, run(nc, tail, g, None, reps=16)), flush=True)
    sys.exit(0)
base = (8, (6, 3), 1.4, None)
print("default chunks %d tail %s growth %.2f fwd tail %s: %.3f ms" % (base + (run(*base),)), flush=True)
res = []
for nc in (6, 8, 10):
    for tail in ((6, 3), (5, 3), (6, 4), (4, 3), (7, 3)):
        for growth in (1.2, 1.3, 1.4, 1.6):
            res.append((run(nc, tail, growth, None), nc, tail, growth, None))
            print("chunks %d tail %s growth %.1f: %.3f ms" % (nc, tail, growth, res[-1][0]), flush=True)
res.sort()
best = res[0]
for tf in ((6, 3), (4, 3), (3, 4), (6, 4), (4, 6)):
    print("best bwd setting %s + fwd tail %s: %.3f ms" % (best[1:4], tf, run(best[1], best[2], best[3], tf)), flush=True)
print("best five:", res[:5])
