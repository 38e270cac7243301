"""cProfile of the HOST side of a train step (enqueue only; the device is synchronised between steps)"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import satt_amd  # noqa: F401
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
ops.set_precision("bf16")
eng = Engine(ModelConfig(), "cuda", rng_seed=3)
b = eng.to_device_batch(synthetic_batch(32, 97, 330, seed=5))
for _ in range(5):
    eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize()
pr = cProfile.Profile()
N = 20
for _ in range(N):
    torch.cuda.synchronize()
    pr.enable(); eng.train_step(b); eng.optimizer_step(); pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
