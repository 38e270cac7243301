"""weight-gradient GEMMs of the small shapes: time against the number of reduction splits (the default comes from ops._splitk)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import satt_amd  # noqa: F401
from satt_amd import ops
ops.set_precision("bf16")
dev = "cuda"
shapes = [(128, 256, 5120, "highway"), (128, 512, 5120, "enc lstm in"), (256, 224, 5120, "keys1"), (128, 128, 5120, "small"),
          (256, 96, 5120, "enc sa kvq"), (256, 256, 12800, "dec o/t"), (128, 1024, 12800, "xg_att"), (256, 768, 12800, "dec kvq")]
for K, N, M, name in shapes:
    x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev)
    dW = torch.zeros(K, N, device=dev); db = torch.zeros(N, device=dev)
    base = ops._splitk(ops._tiles(K, N), M)
    row = []
    for sk in sorted({base, 4, 8, 16, 24, 32, 40, 48, 64, 80}):
        def run():
            ops.gemm(K, N, M, x, K, dy, N, 1, dW, N, a_mode=1, accumulate=True, splitk=sk, colsum=db)
        try:
            for _ in range(5): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run()
            e1.record(); torch.cuda.synchronize()
            row.append("%s%d: %.1f" % ("*" if sk == base else "", sk, e0.elapsed_time(e1) / 50 * 1e3))
        except Exception as e:
            row.append("%d: err" % sk)
    print("%-12s %4dx%4dx%5d  us by splits (* = default)  %s" % (name, K, N, M, "  ".join(row)))
