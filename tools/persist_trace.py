"""diagnostics: wall-clock stamps of member 0 around the phases of one decoder step of the persistent decode kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.inference import infer
ops.set_precision("bf16")
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
g = np.random.default_rng(0)
src = g.integers(1, 68, (1, 100)); sl = np.full((1,), 100, dtype=np.int64)
for _ in range(2):
    out = infer(eng, src, sl, max_steps=64, min_steps=10 ** 6, persistent=True)
ses = eng._decode_sessions[next(reversed(eng._decode_sessions))]
assert ses.persist is not None
G = ses.persist.G
raw = ses.persist_ws.cpu().numpy()
st = np.frombuffer(raw[16 * G + 64:16 * G + 64 + 512].tobytes(), dtype=np.uint64).astype(np.int64)
n = ses.persist.nphase
kinds = [("lin%d" % ses.persist.phase_arg[i]) if ses.persist.phase_kind[i] == 0 else ("energy", "context", "satt")[ses.persist.phase_kind[i] - 1] for i in range(n)]
print("decode %.1f us/step" % (1e3 * out["decode_ms"] / 64))
for i in range(n):
    body = (st[2 * i + 1] - st[2 * i]) * 0.01
    bar = (st[2 * i + 2] - st[2 * i + 1]) * 0.01
    print("%-8s body %6.2f us   barrier %6.2f us" % (kinds[i], body, bar))
print("step %.2f us" % ((st[2 * n] - st[0]) * 0.01))
for name, o in (("lin0 (pre-net 0)", 32), ("lin2 (attention LSTM)", 40)):
    d = [(st[o + i + 1] - st[o + i]) * 0.01 for i in range(5)]
    print("%-22s weights issued %.2f | wait smem %.2f | x staged+barrier %.2f | FMA+red %.2f | barrier %.2f" % (name, *d))
