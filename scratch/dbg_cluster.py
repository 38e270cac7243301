"""compare the cluster attention path with the single-workgroup path, per candidate cluster size"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from common import MEDIUM, SMALL, make_params, small_batch

def run(cfg, P, batch, clusters, sizes):
    ops.ATTN_CLUSTER_SIZES = sizes
    ops.set_precision("f32")
    eng = Engine(cfg, "cuda", params=P, rng_seed=7)
    eng.use_clusters = clusters
    b = eng.to_device_batch(batch)
    eng.zero_grad()
    ctx = eng.forward(b, training=True)
    torch.cuda.synchronize()
    try:
        eng.check_clusters(ctx)
    except Exception as e:
        print("   status:", e)
    return {k: v.detach().float().cpu().numpy() for k, v in eng.outputs(ctx).items()}

for name, kw, B, Ti, Tm in (("SMALL", SMALL, 3, 9, 12), ("MEDIUM", MEDIUM, 5, 37, 46)):
    cfg, P = make_params(kw, seed=1)
    batch = small_batch(cfg, B, Ti, Tm, seed=3)
    ref = run(cfg, P, batch, False, (4,))
    for sizes in ((8,), (4,), (2,)):
        out = run(cfg, P, batch, True, sizes)
        errs = {k: float(np.abs(out[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)) for k in ("alignment1", "alignment2", "dec_out", "mel")}
        print(name, "A=%d" % cfg.att_rnn_units, sizes, {k: "%.2e" % v for k, v in errs.items()})
