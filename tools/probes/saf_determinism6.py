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
for rep in range(3):
    eng = Engine(ModelConfig(), "cuda", param_seed=5, rng_seed=9)
    b = eng.to_device_batch(batch)
    eng.zero_grad(); ctx = eng.forward(b, True)
    torch.cuda.synchronize()
    U = eng.G["dec.att1.U"]
    print("rep", rep, "after forward: nonzero elements of G[dec.att1.U]:", int((U != 0).sum()), " whole grad nonzero:", int((eng.grad != 0).sum()))
    # run the backward with the deferred gradients' outputs redirected: does anything ELSE write G[dec.att1.U]?
    real = ops.attn_param_grads_finish
    ops.attn_param_grads_finish = lambda *a, **k: None
    eng.backward(ctx); torch.cuda.synchronize()
    ops.attn_param_grads_finish = real
    nz = torch.nonzero(U.view(-1) != 0).view(-1).cpu().numpy()
    print("   after backward WITHOUT the finish launch: nonzero elements of G[dec.att1.U]: %d %s values %s" % (len(nz), nz[:20].tolist(), U.view(-1)[nz[:6]].cpu().numpy() if len(nz) else ""))
    eng._pg_acc.zero_()
