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
def d(x, y): return float(np.abs(x - y).max() / (np.abs(y).max() + 1e-30))
for rep in range(4):
    eng = Engine(ModelConfig(), "cuda", param_seed=5, rng_seed=9)
    b = eng.to_device_batch(batch)
    eng.zero_grad(); ctx = eng.forward(b, True); eng.backward(ctx)
    torch.cuda.synchronize(); eng.check_clusters(ctx)
    U_inflight = eng.G["dec.att1.U"].double().cpu().numpy().copy()
    de1, de2 = ctx["_de"]
    c = eng.cfg
    dk1, dk2 = torch.zeros(B * Ti, c.att1_units, device="cuda"), torch.zeros(B * Ti, c.att2_units, device="cuda")
    dv1, db1, dU, dv2 = (torch.zeros(224, device="cuda"), torch.zeros(224, device="cuda"), torch.zeros(5, 224, device="cuda"), torch.zeros(32, device="cuda"))
    ops.attn_param_grads(ctx["att_params"], de1, de2, dk1, dk2, dv1, db1, dU, dv2)
    torch.cuda.synchronize()
    U_off = dU.double().cpu().numpy()
    diff = np.abs(U_inflight - U_off)
    bad = np.argwhere(diff > 0.002 * np.abs(U_off).max())
    print("   differing elements: %d of %d; rows(k) %s; unit range %s..%s; sample %s" % (len(bad), diff.size, sorted(set(bad[:, 0].tolist())),
          bad[:, 1].min() if len(bad) else None, bad[:, 1].max() if len(bad) else None, bad[:6].tolist()))
    print("rep", rep, "in-flight vs offline dU: %.2e" % d(U_inflight, U_off), " offline |dU|max %.3e" % np.abs(U_off).max(),
          " db in-flight vs offline %.2e" % d(eng.G["dec.att1.b"].double().cpu().numpy(), db1.double().cpu().numpy()))
