"""experiment: chunking of the recurrent stream pipeline (Engine.pipeline_chunks / pipeline_tail) vs step time"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
ops.set_precision("bf16")
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
def run(n):
    for _ in range(n):
        eng.train_step(b); eng.optimizer_step()
for chunks, tail in [(6, (3, 4)), (6, (4, 8)), (6, (4, 6)), (7, (4, 8)), (8, (4, 8)), (6, (3, 8)), (6, (5, 16)), (5, (3, 4)), (7, (3, 4))]:
    eng.pipeline_chunks, eng.pipeline_tail = chunks, tail
    run(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(30); torch.cuda.synchronize()
    print("chunks %d tail %s: %.3f ms/step  bounds %s" % (chunks, tail, (time.perf_counter() - t0) / 30 * 1e3, [b1 - b0 for b0, b1 in eng._chunk_bounds(400, chunks)]))
