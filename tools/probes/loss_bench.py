"""us per launch of the fused loss kernel at the benchmark shape (B=32, Td=400, r=2, 80 mels, padded rows)"""
import sys
sys.path.insert(0, '.')
import torch
import satt_amd
from satt_amd import ops
B, Td, r, nm = 32, 400, 2, 80
Tm, NO, NOp = Td * r, r * nm + 1, 168
dev = "cuda"
y = torch.randn(B * Td, NOp, device=dev); dy = torch.empty_like(y)
tgt = torch.randn(B, Tm, nm, device=dev); sm = torch.ones(B, Tm, device=dev); bm = torch.ones(B, Td, device=dev)
done = torch.zeros(B, Td, device=dev); losses = torch.zeros(3, device=dev); ws = torch.zeros(2048, device=dev)
ops.loss_mask_sums(sm, bm, B, Tm, Td, ws)
def run():
    ops.loss_fwd_bwd_presummed(y[:, :NO], NOp, tgt, sm, y[:, NO - 1:], NOp, done, bm, B, Tm, nm, Td, False, losses, dy[:, :NO], NOp,
                               dy[:, NO - 1:], NOp, ws)
for _ in range(10): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print("loss_fused: %.2f us per launch (back to back)" % (e0.elapsed_time(e1) * 1000 / 200))
