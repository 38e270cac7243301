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
vctk = os.environ.get("SATT_MODEL") == "vctk"       # BASELINE config 4's shape
eng = Engine(ModelConfig(num_speakers=152, speaker_offset=225) if vctk else ModelConfig(), "cuda", param_seed=0, rng_seed=1)
for kv in filter(None, os.environ.get("SATT_SET", "").split(";")):       # e.g. SATT_SET=single_launch_attention=False
    k, v = kv.split("=")
    setattr(eng, k, eval(v))
if os.environ.get("SATT_KEEPALL"):        # diagnosis: no temporary of the engine is ever returned to the allocator
    _all, _e0 = [], eng._e
    def _e_keep(*a, **k):
        t = _e0(*a, **k); _all.append(t); return t
    eng._e = _e_keep
b = eng.to_device_batch(synthetic_batch(32, 80, 500, seed=1234, min_source_length=30, min_target_steps=90, num_speakers=152,
                                        speaker_offset=225) if vctk else
                        synthetic_batch(int(os.environ.get("SATT_BATCH", "32")), *(int(x) for x in os.environ.get("SATT_SHAPE", "160,800").split(",")), seed=1234))
NW, NT, NR = (int(x) for x in os.environ.get('SATT_STEPS', '5,20,3').split(','))     # warm-up steps, timed steps, repeats
for _ in range(NW):
    ctx = eng.train_step(b)
torch.cuda.synchronize()
try:
    eng.check_clusters(ctx)
except Exception as e:
    print("WARM-UP:", e)
best = 1e9
for rep in range(NR):
    t0 = time.perf_counter()
    for _ in range(NT):
        ctx = eng.train_step(b)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / NT * 1e3)
try:
    eng.check_clusters(ctx)
except Exception as e:
    print("TIMED:", e)
    Cn, cws1, cws2 = ctx["cluster"]
    for nm, w in (("lstm1", cws1), ("lstm2", cws2)):      # 64-byte tail: error word, exchange-path counters, debug words (SATT_XCHG_DEBUG)
        tail = w[-64:].view(torch.int32).cpu().numpy().astype("uint32")
        print(nm, "tail:", [hex(int(x)) for x in tail[:16]])
print("cluster size %s  ms/step %.3f  loss %.5f" % (ctx["att_cluster"][0], best, float(eng.losses[2])))
