"""real inputs of one engine step; the deferred gradients re-run offline: (a) alone, (b) with a busy neighbour stream"""
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
eng = Engine(ModelConfig(), "cuda", param_seed=5, rng_seed=9)
b = eng.to_device_batch(batch)
eng.zero_grad(); ctx = eng.forward(b, True); eng.backward(ctx)
torch.cuda.synchronize(); eng.check_clusters(ctx)
de1, de2 = ctx["_de"]; ap = ctx["att_params"]; Td = Tm // 2
def offline(busy, pieces, pad):
    dk1, dk2 = torch.zeros(B * Ti, 224, device="cuda"), torch.zeros(B * Ti, 32, device="cuda")
    dv1, db1, dU, dv2 = (torch.zeros(224, device="cuda"), torch.zeros(224, device="cuda"), torch.zeros(5, 224, device="cuda"), torch.zeros(32, device="cuda"))
    acc = ops.attn_param_grads_acc_buffer(ap, "cuda")
    side = torch.cuda.Stream()
    if busy:
        x = torch.randn(4096, 4096, device="cuda")
        with torch.cuda.stream(side):
            for _ in range(30): x = (x @ x) * 1e-4
    first = True
    for (t0, t1) in pieces:
        ops.attn_param_grads_acc(ap, de1, de2, dk1, dk2, acc, t0, t1, accumulate=not first, lds_pad=pad); first = False
    ops.attn_param_grads_finish(ap, acc, dv1, db1, dU, dv2)
    torch.cuda.synchronize()
    return dU.double().cpu().numpy()
def d(x, y): return float(np.abs(x - y).max() / (np.abs(y).max() + 1e-30))
P1 = [(0, Td)]; P8 = [(90, 100), (70, 90), (50, 70), (30, 50), (15, 30), (0, 15)]
ref = offline(False, P1, 0)
for busy in (False, True):
    for pieces, pad in ((P1, 0), (P8, 0), (P8, 96 * 1024)):
        r = [d(offline(busy, pieces, pad), ref) for _ in range(4)]
        print("busy %-5s pieces %d pad %6d: vs single-call reference %s" % (busy, len(pieces), pad, ["%.1e" % v for v in r]))
print("in-flight vs reference %.1e" % d(eng.G["dec.att1.U"].double().cpu().numpy(), ref))
