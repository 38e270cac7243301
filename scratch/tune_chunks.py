"""ms/step for several layer-pipeline chunkings (pipeline_chunks, pipeline_tail)."""
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
for nc, tail in [(6, (3, 4)), (7, (3, 4)), (7, (4, 8)), (8, (4, 8)), (8, (3, 4)), (6, (4, 8)), (6, (3, 6)), (7, (3, 6)), (6, (2, 4)), (6, (3, 4))]:
    eng.pipeline_chunks, eng.pipeline_tail = nc, tail
    run(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(12); torch.cuda.synchronize()
    print(nc, tail, eng._chunk_bounds(400, nc), "%.3f ms" % ((time.perf_counter() - t0) / 12 * 1e3))
