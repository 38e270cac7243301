"""Critical-path phases of a train step from the HIP events the engine records around the named kernels."""
import sys
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
N = 6
marks = []
eng.timing = {}
for i in range(N):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    eng.train_step(b)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    eng.optimizer_step()
    e2 = torch.cuda.Event(enable_timing=True); e2.record()
    marks.append((e0, e1, e2))
torch.cuda.synchronize()
T = eng.timing
per = {k: len(v) // N for k, v in T.items()}
for i in range(2, N):
    e0, e1, e2 = marks[i]
    def rel(ev): return e0.elapsed_time(ev)
    out = {}
    for k, v in T.items():
        evs = v[i * per[k]:(i + 1) * per[k]]
        out[k] = (rel(evs[0][0]), rel(evs[-1][1]))
    print("step %d: total %.2f (bwd end %.2f)" % (i, rel(e2), rel(e1)))
    for k in sorted(out, key=lambda k: out[k][0]):
        print("   %-14s %7.2f -> %7.2f" % (k, out[k][0], out[k][1]))

# where does the tail go: main-stream chain vs the weight-gradient stream
orig_join = eng._wgrad_join
rec = {}
def join():
    rec["main_end"] = torch.cuda.Event(enable_timing=True); rec["main_end"].record()
    if eng._wg_stream is not None:
        rec["wg_end"] = torch.cuda.Event(enable_timing=True); rec["wg_end"].record(eng._wg_stream)
    orig_join()
eng._wgrad_join = join
eng.timing = {}
for i in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    eng.train_step(b)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    eng.optimizer_step()
torch.cuda.synchronize()
ab = eng.timing["attn_rnn_bwd"][-1][1]
print("last step: attn bwd end %.2f  main chain end %.2f  wg stream end %.2f  bwd end %.2f" % (
    e0.elapsed_time(ab), e0.elapsed_time(rec["main_end"]), e0.elapsed_time(rec["wg_end"]), e0.elapsed_time(e1)))
