import sys
sys.path.insert(0, '.')
import torch
import satt_amd
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
for _ in range(4):
    eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize()
for rep in range(2):
    eng.marks = []; eng.timing = {}; eng.timing_names = {"attn_param_grads", "attn_rnn_bwd"}
    eng.train_step(b); eng.optimizer_step()
    torch.cuda.synchronize()
    d = dict(eng.marks)
    a0, a1 = eng.timing["attn_rnn_bwd"][0]
    print("attention bwd: start %+.3f  dur %.3f (relative to head-bwd mark)" % (d["decoder head bwd"].elapsed_time(a0), a0.elapsed_time(a1)))
    for (x, y) in eng.timing["attn_param_grads"]:
        print("  PG start %+.3f dur %.3f   (end %+.3f vs attention end)" % (a0.elapsed_time(x), x.elapsed_time(y), a1.elapsed_time(y)))
