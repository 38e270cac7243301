#!/usr/bin/env python
"""The conv bank's forward alone (B = 32, Ti = 160, 16 widths, 128 -> 128 channels): HIP-event time per launch.
`python tools/bench_conv_bank.py [iters]`; under rocprofv3 --pmc for the kernel's cache counters."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import satt_amd  # noqa: E402,F401
from satt_amd import ops  # noqa: E402

DEV = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B, Ti = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 160)
ops.set_precision("bf16")
g = torch.Generator().manual_seed(0)
M = B * Ti
x = torch.randn(M, 128, generator=g).to(DEV)
Ws = [torch.randn(k, 128, 128, generator=g) / math.sqrt(k * 128) for k in range(1, 17)]
flat = torch.cat([w.reshape(-1) for w in Ws]).contiguous().to(DEV)
st = torch.zeros(flat.numel(), dtype=torch.bfloat16, device=DEV); sn = torch.zeros_like(st)
tab, off = [], 0
for k in range(1, 17):
    tab += [off, k, 128, 128]; off += k * 128 * 128
ops.shadow_pack(flat, torch.tensor(tab, dtype=torch.int64, device=DEV), 16, st, sn)
Wb = ops.Weight(flat, st, sn)
out = torch.empty(M, 2048, device=DEV)
for _ in range(5):
    ops.conv_bank(x, Ti, Wb, 16, out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    ops.conv_bank(x, Ti, Wb, 16, out)
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3 / iters
f = 2.0 * M * 128 * 128 * 136
print("conv bank fwd B=%d Ti=%d: %.1f us per launch, %.1f TFLOP/s (%.3f of 2500)" % (B, Ti, us, f / us / 1e6, f / us / 1e6 / 2500))

