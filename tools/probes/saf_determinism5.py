"""diagnosis build (SATT_PG_CHECKSUM): sums of what the deferred-gradient kernel READ in flight vs offline"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
ops.set_precision("bf16")
B, Ti, Tm = 3, 160, 200
batch = synthetic_batch(B, Ti, Tm, seed=77)
for rep in range(4):
    eng = Engine(ModelConfig(), "cuda", param_seed=5, rng_seed=9)
    b = eng.to_device_batch(batch)
    eng.zero_grad(); ctx = eng.forward(b, True); eng.backward(ctx)
    torch.cuda.synchronize()
    n0 = 7 * 224 + 32
    inflight = eng._pg_acc[n0:n0 + 4].cpu().numpy().copy(); wg_in = eng._pg_acc[n0 + 8:n0 + 8 + 240].cpu().numpy().copy(); eng._pg_acc[n0:].zero_()
    de1, de2 = ctx["_de"]; ap = ctx["att_params"]
    dk1, dk2 = torch.zeros(B * Ti, 224, device="cuda"), torch.zeros(B * Ti, 32, device="cuda")
    acc = ops.attn_param_grads_acc_buffer(ap, "cuda")
    ops.attn_param_grads_acc(ap, de1, de2, dk1, dk2, acc)
    torch.cuda.synchronize()
    off = acc[n0:n0 + 4].cpu().numpy(); wg_off = acc[n0 + 8:n0 + 8 + 240].cpu().numpy()
    dd = (wg_in - wg_off).reshape(120, 2)
    badwg = np.nonzero(np.abs(dd[:, 0]) > 1e-12)[0]
    print("   per-workgroup partial of dU[1][192]: %d of 120 differ: %s ; dU[1][193]: %d differ" % (len(badwg), [(int(w) // 40, int(w) % 40, "%.2e" % dd[w, 0], "%.2e" % wg_off.reshape(120, 2)[w, 0]) for w in badwg[:6]], int((np.abs(dd[:, 1]) > 1e-12).sum())))
    print("rep", rep, "in-flight - offline sums [saf even, saf odd, de, fl1]:", ["%.3e" % x for x in (inflight - off)], " offline", ["%.6e" % x for x in off])
