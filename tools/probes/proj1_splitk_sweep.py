"""proj1 (conv k=3, 2048 -> 128 channels, B*Ti = 5120 rows) forward: split-K sweep on gemm_rk_k (SATT_TILE_BM=64|128 outside)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import satt_amd
from satt_amd import ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_gemm import weight
ops.set_precision("bf16")
DEV = "cuda"
g = torch.Generator().manual_seed(0)
B, Ti = 32, 160; M = B * Ti
mp = torch.randn(M, 2048, generator=g).to(DEV)
W1 = (torch.randn(3, 2048, 128, generator=g) / math.sqrt(6144)).to(DEV)
Ww = weight(W1)
out = torch.empty(M, 128, device=DEV)
W, Wt = Ww.w, Ww.t
def run(sk):
    ops.gemm(M, 128, 3 * 2048, mp, 2048, W, 128, 1, out, 128, a_mode=2, conv=(Ti, 2048, 1, -1), kin=2048, sb_tap=2048 * 128,
             splitk=sk, split_overwrite=True, Bs=Wt, sbs_tap=2048 * 128, sbs_n=2048)
for sk in (1, 2, 3, 4, 6, 8, 12, 16, 24):
    for _ in range(3): run(sk)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): run(sk)
    b.record(); torch.cuda.synchronize()
    print("splitk %2d: %.1f us" % (sk, a.elapsed_time(b) * 1e3 / 30))
