"""How long does the HOST take to enqueue one train step (no device sync inside)?"""
import sys, time, torch
sys.path.insert(0, '.')
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
ops.set_precision("bf16")
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
for _ in range(3):
    eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize()
for timing in (None, {}):
    eng.timing = timing
    enq = []
    t00 = time.perf_counter()
    for _ in range(10):
        t0 = time.perf_counter()
        eng.train_step(b); eng.optimizer_step()
        enq.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t00
    print("timing=%s: host enqueue per step (ms): %s | wall per step %.2f ms" % (timing is not None, " ".join("%.1f" % (1e3 * x) for x in enq), 1e3 * tot / 10))
import cProfile, pstats
eng.timing = None
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    eng.train_step(b); eng.optimizer_step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
