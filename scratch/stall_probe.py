import sys, time, os
sys.path.insert(0, '.')
import torch
import satt_amd
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
mode = sys.argv[1]
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
for _ in range(5):
    eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize()
N = 20
if mode in ("timing", "pregrow"):
    eng.timing = {}
if mode == "pregrow":
    warm = [torch.cuda.Event(enable_timing=True) for _ in range(80 * N)]
    for e in warm: e.record()
    torch.cuda.synchronize(); del warm
marks = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
marks[0].record()
host = []
for i in range(N):
    t0 = time.perf_counter()
    eng.train_step(b); eng.optimizer_step()
    marks[i + 1].record()
    host.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
print(mode, "gpu :", " ".join("%.1f" % marks[i].elapsed_time(marks[i + 1]) for i in range(N)))
print(mode, "host:", " ".join("%.1f" % x for x in host))
