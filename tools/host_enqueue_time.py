#!/usr/bin/env python
"""Host time to ENQUEUE one train step (no synchronisation inside the step): the host must stay ahead of the GPU (9.6 ms per
step) for the dependent chain not to starve.  --dist: with a one-rank RCCL process group carrying the two gradient buckets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import satt_amd  # noqa: F401
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.parallel import DataParallel
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch

dist_on = "--dist" in sys.argv
ops.set_precision("bf16")
cfg = ModelConfig()
dp = DataParallel(1, 0, 0, backend="nccl", force=dist_on)
eng = Engine(cfg, "cuda", rng_seed=3)
dp.bind(eng.grad)
_B, _Ti, _Tm = (int(v) for v in os.environ.get("SATT_BTT", "32,160,800").split(","))      # SATT_BTT=B,Ti,Tm
b = eng.to_device_batch(synthetic_batch(_B, _Ti, _Tm, num_mels=cfg.num_mels, r=cfg.r, seed=5))
ar = dp.allreduce if dp.active else None
for _ in range(5):
    eng.train_step(b, allreduce=ar); dp.wait(); eng.optimizer_step()
torch.cuda.synchronize()
tf, tb, tw, to = [], [], [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.zero_grad(); ctx = eng.forward(b, training=True); t1 = time.perf_counter()
    if ar is not None:
        eng.backward(ctx, on_decoder_grads_ready=lambda: ar(eng.enc_end, eng.nparam)); ar(0, eng.enc_end)
    else:
        eng.backward(ctx)
    t2 = time.perf_counter(); dp.wait(); t3 = time.perf_counter(); eng.optimizer_step(); t4 = time.perf_counter()
    tf.append(t1 - t0); tb.append(t2 - t1); tw.append(t3 - t2); to.append(t4 - t3)
m = lambda v: 1e3 * float(np.median(v))
print("host enqueue ms%s: forward %.2f backward %.2f wait %.2f optimizer %.2f total %.2f"
      % (" (one-rank RCCL)" if dist_on else "", m(tf), m(tb), m(tw), m(to), m(tf) + m(tb) + m(tw) + m(to)))
dp.shutdown()
