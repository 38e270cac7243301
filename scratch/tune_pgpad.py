import sys, time
sys.path.insert(0, '.')
import torch
import satt_amd
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
def run(n):
    for _ in range(n):
        eng.train_step(b); eng.optimizer_step()
run(3); torch.cuda.synchronize()
for pad in [96, 64, 48, 40, 32, 24, 8, 96]:
    eng.pg_lds_pad = pad * 1024
    run(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(20); torch.cuda.synchronize()
    print("pad %3d KB: %.3f ms/step" % (pad, (time.perf_counter() - t0) / 20 * 1e3))
