import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import satt_amd
from satt_amd import ops
ops.set_precision("bf16")
DEV="cuda"
g = torch.Generator().manual_seed(0)
B, Ti = 32, 160; M = B*Ti
x = torch.randn(M, 128, generator=g).to(DEV); dy = torch.randn(M, 2048, generator=g).to(DEV)
nw = 136*128*128
dW = torch.zeros(nw, device=DEV)
def run(sk):
    return ops.gemm(16 * 128, 128, M, x, 128, dy, 2048, 1, dW, 128, a_mode=3, conv=(Ti, 128, 1, 0), accumulate=True,
            splitk=sk, bank=(16, 0, 128, 128 * 128), only_path=2)
for sk in (1, 2, 3, 4, 5, 6, 8, 12, 16):
    for _ in range(3): run(sk)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): run(sk)
    b.record(); torch.cuda.synchronize()
    print("splitk %2d: %.1f us" % (sk, a.elapsed_time(b) * 1e3 / 30))
