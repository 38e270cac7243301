"""ms per train step of the benchmark workload under experiment switches (SATT_CMAX: attention cluster sizes to try,
SATT_LIB_PATH: a variant library); same-box A/B helper, not the benchmark (bench.py is)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
if os.environ.get("SATT_CMAX"):
    ops.ATTN_CLUSTER_SIZES = tuple(int(x) for x in os.environ["SATT_CMAX"].split(","))
if os.environ.get("SATT_NO_FUSED_BN"):       # A/B: three-launch BatchNorm everywhere
    ops.bn_fwd_fused = lambda *a, **k: False
    ops.bn_bwd_fused = lambda *a, **k: False
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
for _ in range(5):
    ctx = eng.train_step(b)
torch.cuda.synchronize()
eng.check_clusters(ctx)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        ctx = eng.train_step(b)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
eng.check_clusters(ctx)
print("cluster size %s  ms/step %.3f  loss %.5f" % (ctx["att_cluster"][0], best, float(eng.losses[2])))
