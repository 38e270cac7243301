"""diagnostic: decoder-step time with the GPU otherwise idle vs with a dense GEMM stream keeping the clocks up"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import satt_amd  # noqa
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.inference import infer
ops.set_precision("bf16")
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
g = np.random.default_rng(0)
src = g.integers(1, 68, (1, 100)); sl = np.full((1,), 100, dtype=np.int64)
kw = dict(max_steps=200, min_steps=10 ** 6)
infer(eng, src, sl, **kw)
for rep in range(3):
    out = infer(eng, src, sl, **kw)
    print("idle   : %.1f us/step" % (1e3 * out["decode_ms"] / 200))
print(subprocess.run("rocm-smi --showclocks 2>/dev/null | grep -i 'sclk\\|mclk' | head -4", shell=True, capture_output=True, text=True).stdout)
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16); b = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for rep in range(3):
    with torch.cuda.stream(side):
        for _ in range(400):
            c = a @ b
    out = infer(eng, src, sl, **kw)
    torch.cuda.synchronize()
    print("busy   : %.1f us/step" % (1e3 * out["decode_ms"] / 200))
