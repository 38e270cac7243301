import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
ops.set_precision("bf16")
cfg = ModelConfig()
eng = Engine(cfg, "cuda", rng_seed=3)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, num_mels=cfg.num_mels, r=cfg.r, seed=5))
for _ in range(5):
    eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize()
tf, tb, to = [], [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.zero_grad(); ctx = eng.forward(b, training=True); t1 = time.perf_counter()
    eng.backward(ctx); t2 = time.perf_counter(); eng.optimizer_step(); t3 = time.perf_counter()
    tf.append(t1 - t0); tb.append(t2 - t1); to.append(t3 - t2)
print("host enqueue ms: forward %.2f backward %.2f optimizer %.2f total %.2f" % (1e3*np.median(tf), 1e3*np.median(tb), 1e3*np.median(to), 1e3*(np.median(tf)+np.median(tb)+np.median(to))))
