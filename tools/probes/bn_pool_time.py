"""BatchNorm + ReLU + max-pool of the conv bank (B*Ti = 5120 rows x 2048 channels): forward and backward pair, HIP-event time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import satt_amd
from satt_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(0)
B, Tn, Cc = 32, 160, 2048
rows = B * Tn
x = (torch.randn(rows, Cc, generator=g) * 2 + 0.3).to(DEV)
gamma, beta = torch.randn(Cc, generator=g).to(DEV), torch.randn(Cc, generator=g).to(DEV)
dmp = torch.randn(rows, Cc, generator=g).to(DEV)
mp = torch.empty_like(x); mean, rstd = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
mm, mv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
ws = ops.bn_ws(rows, Cc, DEV)
dx = torch.empty_like(x); dg, db = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV); dbuf = torch.empty_like(x)
def fwd(): ops.bn_maxpool_fwd(x, gamma, beta, mp, mean, rstd, mm, mv, ws, B, Tn, 1e-3, 0.99, ops.ACT_RELU)
def bwd(): ops.maxpool_bn_bwd(dmp, x, gamma, beta, mean, rstd, dx, dg, db, ws, dbuf, B, Tn, ops.ACT_RELU)
for name, fn in (("forward (3 launches)", fwd), ("backward (3 launches)", bwd)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): fn()
    b.record(); torch.cuda.synchronize()
    print("%s: %.1f us" % (name, a.elapsed_time(b) * 1e3 / 30))
