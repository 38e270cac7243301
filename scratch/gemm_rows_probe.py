import sys
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import satt_amd
from satt_amd import ops
dev = "cuda"
B, Td, D, G4 = 32, 400, 256, 1024
dxg = torch.randn(B * Td, G4, device=dev)
W = torch.randn(D, G4, device=dev) * 0.05          # [K=256 rows(out), N=1024]: dx = dy @ W^T
W1 = torch.randn(544, G4, device=dev) * 0.05
dh1 = torch.zeros(B * Td, D, device=dev)
datt = torch.zeros(B * Td, 544, device=dev)
xg = torch.zeros(B * Td, G4, device=dev)
att_out = torch.randn(B * Td, 544, device=dev)
bias = torch.zeros(G4, device=dev)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (t0, t1) in [(384, 400), (352, 384), (288, 352), (192, 288)]:
    print("rows %3d: dx256 %.1f us  dx544 %.1f us  fwd(544->1024) %.1f us  fwd(256->1024) %.1f us" % (
        t1 - t0,
        timeit(lambda: ops.linear_dx_rows(dxg, W, dh1, B, Td, t0, t1)),
        timeit(lambda: ops.linear_dx_rows(dxg, W1, datt, B, Td, t0, t1)),
        timeit(lambda: ops.linear_rows(att_out, W1, bias, xg, B, Td, t0, t1)),
        timeit(lambda: ops.linear_rows(dh1, W, bias, xg, B, Td, t0, t1))))
print("padded leading dimensions (+32 floats):")
Wp = torch.randn(D, G4 + 32, device=dev)[:, :G4]
dxgp = torch.randn(B * Td, G4 + 32, device=dev)[:, :G4]
for (t0, t1) in [(384, 400), (288, 352)]:
    print("rows %3d: dx256 W padded %.1f us   A padded %.1f us   both %.1f us   none %.1f us" % (
        t1 - t0,
        timeit(lambda: ops.linear_dx_rows(dxg, Wp, dh1, B, Td, t0, t1)),
        timeit(lambda: ops.linear_dx_rows(dxgp, W, dh1, B, Td, t0, t1)),
        timeit(lambda: ops.linear_dx_rows(dxgp, Wp, dh1, B, Td, t0, t1)),
        timeit(lambda: ops.linear_dx_rows(dxg, W, dh1, B, Td, t0, t1))))
