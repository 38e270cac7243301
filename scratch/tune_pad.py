import sys, time, re
sys.path.insert(0, '.')
import torch
import satt_amd
from satt_amd import engine as E, ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
orig = ops.attn_param_grads
pad = [96 * 1024]
def patched(*a, **kw):
    if "lds_pad" in kw:
        kw["lds_pad"] = pad[0]
    return orig(*a, **kw)
ops.attn_param_grads = patched
def run(n):
    for _ in range(n):
        eng.train_step(b); eng.optimizer_step()
run(3); torch.cuda.synchronize()
for pd in (96, 64, 48, 32, 0, 96, 48):
    pad[0] = pd * 1024
    run(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(30); torch.cuda.synchronize()
    print("pad %3d KB: %.3f ms" % (pd, (time.perf_counter() - t0) / 30 * 1e3))
