"""which hand-off times out on the VCTK workload (B=32, Ti<=80, Tm<=500)?  python tools/probes/vctk_flaky.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
if os.environ.get("SATT_NO_FUSED_BN"):
    ops.bn_fwd_fused = lambda *a, **k: False
    ops.bn_bwd_fused = lambda *a, **k: False
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = ModelConfig(num_speakers=152, speaker_offset=225)
batch = synthetic_batch(32, 80, 500, seed=4321, min_source_length=30, min_target_steps=90, num_speakers=152, speaker_offset=225)
bad = 0
for rep in range(reps):
    for prec in ("bf16", "f32"):
        ops.set_precision(prec)
        eng = Engine(cfg, "cuda", param_seed=3, rng_seed=5)
        for kv in filter(None, os.environ.get("SATT_SET", "").split(";")):       # e.g. SATT_SET=single_launch_attention=False
            k, v = kv.split("=")
            setattr(eng, k, eval(v))
        b = eng.to_device_batch(batch)
        for it in range(2):
            eng.zero_grad()
            t0 = time.perf_counter()
            ctx = eng.forward(b, True)
            t1 = time.perf_counter()
            if os.environ.get("SYNC_MID"):
                torch.cuda.synchronize(); t1 = time.perf_counter()
                try:
                    eng.check_clusters(ctx)
                except Exception as e:
                    bad += 1; print("rep", rep, prec, "iter", it, "after FORWARD (%.3f s):" % (t1 - t0), e); break
            eng.backward(ctx)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            try:
                eng.check_clusters(ctx)
            except Exception as e:
                Cn, cws1, cws2 = ctx["cluster"]
                which = []
                for nm, w in (("lstm1", cws1), ("lstm2", cws2)):
                    try:
                        ops.lstm_cluster_status(w, 32, cfg.dec_units, Cn)
                    except Exception:
                        which.append(nm)
                bad += 1; print("rep", rep, prec, "iter", it, "after BACKWARD (%.3f s):" % (t2 - t1), e, which,
                                "fwd %.3f s" % (t1 - t0)); break
        del eng
ops.set_precision("bf16")
print("failures:", bad, "of", reps * 2)
