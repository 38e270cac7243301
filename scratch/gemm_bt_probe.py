import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import satt_amd
from satt_amd import ops
dev = "cuda"
for (M, K, N) in [(12800, 256, 1024), (5120, 128, 256), (12800, 544, 1024), (12800, 256, 768), (5120, 256, 128), (12800, 160, 256)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) * 0.05; WT = W.t().contiguous()
    b = torch.zeros(N, device=dev); y1 = torch.empty(M, N, device=dev); y2 = torch.empty(M, N, device=dev)
    for _ in range(20):
        ops.gemm(M, N, K, x, K, W, N, 1, y1, N, bias=b)          # B n-contiguous (W [K][N])
    for _ in range(20):
        ops.gemm(M, N, K, x, K, WT, 1, K, y2, N, bias=b)         # B k-contiguous (W^T [N][K])
    torch.cuda.synchronize()
    print(M, K, N, "max diff", float((y1 - y2).abs().max()))
