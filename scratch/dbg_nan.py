import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from common import MEDIUM, SMALL, make_params, small_batch
torch.set_printoptions(linewidth=200, precision=4)
cfg, P = make_params(SMALL, seed=1)
batch = small_batch(cfg, 3, 9, 12, seed=3)
res = {}
for clusters in (False, True):
    ops.ATTN_CLUSTER_SIZES = (2,)
    ops.set_precision("f32")
    eng = Engine(cfg, "cuda", params=P, rng_seed=7)
    eng.use_clusters = clusters
    b = eng.to_device_batch(batch)
    ctx = eng.forward(b, training=True)
    torch.cuda.synchronize()
    res[clusters] = {k: ctx[k].detach().float().cpu() for k in ("al1", "al2", "a1", "att_out")}
    res[clusters]["saved"] = [x.detach().float().cpu() for x in ctx["att_saved"]]
    print("clusters", clusters, "C", ctx["att_cluster"][0])
for k in ("al1", "al2", "a1", "att_out"):
    r, o = res[False][k], res[True][k]
    print(k, "shape", tuple(o.shape), "nan count", int(torch.isnan(o).sum()), "maxdiff", float((torch.nan_to_num(o) - r).abs().max()))
o, r = res[True]["att_out"], res[False]["att_out"]
o = o.view(3, -1, o.shape[-1]); r = r.view(3, -1, r.shape[-1])
print("att_out sample0 step0 cluster:", o[0, 0])
print("att_out sample0 step0 ref    :", r[0, 0])
print("al1 s0 t0 cluster:", res[True]["al1"].view(3, -1, 9)[0, 0])
print("al1 s0 t0 ref    :", res[False]["al1"].view(3, -1, 9)[0, 0])
print("a1 s0 t0 cluster:", res[True]["a1"].view(3, -1, 9)[0, 0])
print("a1 s0 t0 ref    :", res[False]["a1"].view(3, -1, 9)[0, 0])
print("al2 s0 t0 cluster:", res[True]["al2"].view(3, -1, 9)[0, 0])
print("al2 s0 t0 ref    :", res[False]["al2"].view(3, -1, 9)[0, 0])
