"""attn_param_grads_saf_k in isolation: determinism + a float64 reference of dU / dv / db / dkeys from synthetic inputs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import satt_amd
from satt_amd import ops
B, Td, Ti, U1, U2, F = 3, 24, 160, 224, 32, 5
UQ = U1 + U2
g = torch.Generator().manual_seed(1)
dev = "cuda"
R = lambda *s: torch.randn(*s, generator=g).to(dev)
lengths = torch.tensor([160, 131, 97], dtype=torch.int64, device=dev)
fl = R(B, Td, Ti, F); s = (torch.rand(B, Td, Ti, UQ, generator=g) - 0.5).to(dev).to(torch.float16)
de1 = R(B, Td, Ti) * 0.1; de2 = R(B, Td, Ti) * 0.1
v1, v2, b1 = R(U1), R(U2), R(U1)
keys1, keys2 = R(B, Ti, U1), R(B, Ti, U2)
pq = R(B, Td, UQ); locU = R(F, U1)
ap = ops.attn_rnn_params(B=B, Td=Td, Ti=Ti, A=256, U1=U1, V1=256, U2=U2, V2=32, kernel=10, filters=F, training=1, keys_lds_bf16=1,
                         lengths=lengths, keys1=keys1, keys2=keys2, v1=v1, v2=v2, b1=b1, locU=locU, pq=pq, fl=fl, saf=s)
def run(pad):
    dk1, dk2 = torch.zeros(B, Ti, U1, device=dev), torch.zeros(B, Ti, U2, device=dev)
    dv1, db1, dU, dv2 = torch.zeros(U1, device=dev), torch.zeros(U1, device=dev), torch.zeros(F, U1, device=dev), torch.zeros(U2, device=dev)
    ops.attn_param_grads(ap, de1, de2, dk1, dk2, dv1, db1, dU, dv2, 0, 10, accumulate=False, lds_pad=pad)
    ops.attn_param_grads(ap, de1, de2, dk1, dk2, dv1, db1, dU, dv2, 10, Td, accumulate=True, lds_pad=pad)
    torch.cuda.synchronize()
    return [x.double().cpu().numpy() for x in (dk1, dk2, dv1, db1, dU, dv2)]
# float64 reference
sf = s.double().cpu().numpy(); th = -2 * sf
m = (np.arange(Ti)[None, :] < lengths.cpu().numpy()[:, None]).astype(np.float64)          # [B,Ti]
d1 = de1.double().cpu().numpy() * m[:, None, :]; d2 = de2.double().cpu().numpy() * m[:, None, :]
g1 = d1[..., None] * v1.double().cpu().numpy() * (1 - th[..., :U1] ** 2)
g2 = d2[..., None] * v2.double().cpu().numpy() * (1 - th[..., U1:] ** 2)
ref = [g1.sum(1), g2.sum(1), (d1[..., None] * th[..., :U1]).sum((0, 1, 2)), g1.sum((0, 1, 2)),
       np.einsum("btik,btiu->ku", fl.double().cpu().numpy(), g1), (d2[..., None] * th[..., U1:]).sum((0, 1, 2))]
names = ["dkeys1", "dkeys2", "dv1", "db1", "dU", "dv2"]
for pad in (0, 96 * 1024):
    a, b = run(pad), run(pad)
    for n, x, y, r in zip(names, a, b, ref):
        print("pad %6d %-7s run-to-run %.2e   vs float64 %.2e" % (pad, n, np.abs(x - y).max() / (np.abs(r).max() + 1e-30), np.abs(x - r).max() / (np.abs(r).max() + 1e-30)))
