"""Which kernel family (satt_gemm_path: 0 generic, 1 large-tile forward / dX, 2 large-tile dW, 3 conv bank forward) every GEMM of one train step
runs on, with its shape.  `python tools/gemm_paths.py` on the GPU box."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import satt_amd  # noqa: E402,F401
from satt_amd import _lib, ops  # noqa: E402
from satt_amd.datasets.synthetic import synthetic_batch  # noqa: E402
from satt_amd.engine import Engine  # noqa: E402
from satt_amd.params import ModelConfig  # noqa: E402

eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize()
rows = []
orig = _lib.lib().satt_gemm


def spy(pref, stream):
    p = pref._obj
    rows.append((_lib.lib().satt_gemm_path(pref), p.a_mode, p.M, p.N, p.K, p.nb_outer * p.nb_inner, p.splitk, p.bank_ng))
    return orig(pref, stream)


_lib.lib().satt_gemm = spy
eng.train_step(b)
torch.cuda.synchronize()
cnt = collections.Counter(rows)
print("path a_mode      M      N      K batch splitk bank  calls")
for (k, n) in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[0][2] * kv[0][3] * kv[0][4])):
    print("%4d %6d %6d %6d %6d %5d %6d %4d %6d" % (k + (n,)))
print("calls per path:", dict(collections.Counter(r[0] for r in rows)))
