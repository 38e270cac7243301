"""Does the host run ahead of the GPU?  Host return time of every train step vs the GPU completion time of that step."""
import sys, time
sys.path.insert(0, '.')
import torch
import satt_amd
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
for _ in range(3):
    eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize()
N = 8
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
evs[0].record()
t0 = time.perf_counter()
for i in range(N):
    eng.train_step(b); eng.optimizer_step()
    evs[i + 1].record()
    host.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
gpu = [evs[0].elapsed_time(evs[i + 1]) for i in range(N)]
print("host return (ms):", [round(x, 2) for x in host])
print("gpu  done   (ms):", [round(x, 2) for x in gpu])
# phases of the host inside one step
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
eng.train_step(b); eng.optimizer_step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
