import sys, time, gc
sys.path.insert(0, '.')
import torch
import satt_amd
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
log = []
def cb(phase, info):
    log.append((time.perf_counter(), phase, info["generation"], info.get("collected", 0)))
gc.callbacks.append(cb)
for _ in range(3):
    eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize()
N = 30
host = []
for i in range(N):
    t0 = time.perf_counter()
    eng.train_step(b); eng.optimizer_step()
    host.append((t0, time.perf_counter()))
torch.cuda.synchronize()
print("host ms per step:", " ".join("%.1f" % (1e3 * (b_ - a)) for a, b_ in host))
st = None
for t, ph, gen, col in log:
    if ph == "start": st = t
    else:
        if gen >= 1 or (t - st) > 1e-3:
            k = [i for i, (a, b_) in enumerate(host) if a <= t <= b_]
            print("gc gen%d %.2f ms collected %d in step %s" % (gen, 1e3 * (t - st), col, k))
print(gc.get_count(), gc.get_threshold(), len(gc.get_objects()))
