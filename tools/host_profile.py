#!/usr/bin/env python
"""cProfile of the HOST side of train steps (enqueue only: one synchronisation around N steps): where the Python time per step goes.
usage: python tools/host_profile.py [N=20] [top=40]"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satt_amd  # noqa: F401
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ops.set_precision("bf16")
cfg = ModelConfig()
eng = Engine(cfg, "cuda", rng_seed=3)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, num_mels=cfg.num_mels, r=cfg.r, seed=5))
for _ in range(5):
    eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    eng.train_step(b); eng.optimizer_step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).sort_stats(key)
    st.print_stats(top)
    txt = s.getvalue()
    print("==== by %s (times are for %d steps: divide by %d) ====" % (key, N, N))
    print("\n".join(l[:170] for l in txt.splitlines()[4:]))
