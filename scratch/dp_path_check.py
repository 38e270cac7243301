"""Exercise the data-parallel code path of Engine.train_step on ONE GPU: the bucket callbacks (decoder bucket issued
from the weight-gradient stream, encoder bucket at the end) with a stand-in all-reduce (x2 then /2 on the gradient
slice, on the stream the callback runs on) must leave gradients and the update identical to the plain path."""
import sys
sys.path.insert(0, '.')
import torch
import satt_amd
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch

def run(use_cb):
    eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
    b = eng.to_device_batch(synthetic_batch(4, 64, 96, seed=7, min_source_length=30, min_target_steps=20))
    calls = []
    def ar(lo, hi):
        calls.append((lo, hi, torch.cuda.current_stream().cuda_stream))
        eng.grad[lo:hi].mul_(2.0); eng.grad[lo:hi].mul_(0.5)
    for _ in range(2):
        eng.train_step(b, allreduce=ar if use_cb else None)
        eng.optimizer_step()
    torch.cuda.synchronize()
    return eng.grad.clone(), eng.flat.clone(), float(eng.losses[2]), calls

g0, p0, l0, _ = run(False)
g1, p1, l1, calls = run(True)
print("buckets:", [(lo, hi) for lo, hi, _ in calls[-2:]], "streams differ:", calls[-2][2] != calls[-1][2])
print("loss", l0, l1, "max |dgrad|", float((g0 - g1).abs().max()), "max |dparam|", float((p0 - p1).abs().max()))
assert abs(l0 - l1) < 1e-4 and float((g0 - g1).abs().max()) < 1e-3 * float(g0.abs().max())
print("dp path ok")
