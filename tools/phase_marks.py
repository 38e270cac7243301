"""Main-stream phase boundaries of one train step (HIP events, no profiler).
`--dist`: the same with a ONE-RANK RCCL process group carrying the two gradient buckets (parallel.DataParallel(force=True))."""
import os
import sys
sys.path.insert(0, '.')
import torch
import satt_amd
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
from satt_amd.parallel import DataParallel
DIST = "--dist" in sys.argv
if DIST:
    os.environ.setdefault("MASTER_PORT", "29533")
dp = DataParallel(1, 0, 0, backend="nccl", force=DIST)
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
for kv in filter(None, os.environ.get("SATT_SET", "").split(";")):       # A/B switches: SATT_SET="attr=value;..."
    k, v = kv.split("=")
    setattr(eng, k, eval(v))
dp.bind(eng.grad)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
AR = dp.allreduce if DIST else None
for _ in range(4):
    eng.train_step(b, allreduce=AR); dp.wait(); eng.optimizer_step()
torch.cuda.synchronize()
acc = {}
side = {}
N = 5
prev_end, gaps = None, []
for i in range(N):
    eng.marks = []
    eng.side_marks = []
    eng.train_step(b, allreduce=AR)
    dp.wait()
    eng.optimizer_step()
    eng._mark("optimizer")
    torch.cuda.synchronize()
    m = eng.marks
    if prev_end is not None:
        gaps.append(prev_end.elapsed_time(m[0][1]))
    prev_end = m[-1][1]
    for (n0, e0), (n1, e1) in zip(m[:-1], m[1:]):
        acc.setdefault(n1, []).append(e0.elapsed_time(e1))
    for n1, e1 in eng.side_marks:
        side.setdefault(n1, []).append(dict(m)["loss"].elapsed_time(e1))
tot = 0
for k, v in acc.items():
    x = sum(v) / len(v); tot += x
    print("%-30s %7.3f ms" % (k, x))
print("%-30s %7.3f ms" % ("total", tot))
print("main stream, end of the optimizer of step n -> `fwd start` of step n + 1: %.3f ms (with a host synchronisation between the steps)" % (sum(gaps) / len(gaps)))
print("offsets from the `loss` mark (events on the streams named):")
for k, v in side.items():
    print("  %-78s %+7.3f ms" % (k, sum(v) / len(v)))
d = dict(eng.marks)
if getattr(eng, "_pg_mark", None) is not None and "decoder loop bwd" in d:
    print("last deferred attention gradients end %+.3f ms after the loop mark; memory-gradient mark %+.3f" % (
        d["decoder loop bwd"].elapsed_time(eng._pg_mark), d["decoder loop bwd"].elapsed_time(d["memory gradients"])))
