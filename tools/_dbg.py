import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from test_inference_gpu import make_params, small_batch, make_engine, rel_err
from satt_amd.inference import infer, DecodeSession
for B, Ti, steps in [(1, 57, 19), (2, 100, 19), (2, 57, 19), (2, 57, 3)]:
    cfg, P = make_params(dict(), seed=4)
    batch = small_batch(cfg, B, Ti, steps * cfg.r, seed=6)
    eng, _ = make_engine(cfg, dict(P), "bf16")
    kw = dict(max_steps=steps, min_steps=10 ** 6)
    DecodeSession.MEGA = True; DecodeSession.MEGA_MAX_B = 4
    new = infer(eng, batch["source"], batch["source_length"], **kw)
    DecodeSession.MEGA = False
    old = infer(eng, batch["source"], batch["source_length"], **kw)
    DecodeSession.MEGA = True
    print(B, Ti, steps, "lengths", np.asarray(batch["source_length"]))
    for k in ("mel", "stop", "alignment1", "alignment2"):
        for b in range(B):
            a, o = new[k][b].cpu().numpy(), old[k][b].cpu().numpy()
            per_t = [rel_err(a[t:t + 1], o[t:t + 1]) for t in range(min(4, a.shape[0]))]
            print("  ", k, "b", b, "err %.2e" % rel_err(a, o), "first steps", ["%.1e" % x for x in per_t])
