"""steady-state (unsynchronised) train steps over model variants and precisions: hand-off timeouts, finite losses, ms per step"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import satt_amd
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
base_tac = dict(sa_units=0, att2_units=0, dec_sa_units=0, att1_units=256)
variants = {
    "default": {}, "baseline tacotron": base_tac, "baseline + l2": dict(base_tac, l2_weight=1e-6),
    "transition agent": dict(transition_agent=True), "location_sensitive": dict(attention="location_sensitive"),
    "cumulative": dict(cumulative_weights=True), "postnet v2": dict(use_postnet_v2=True),
    "vctk": dict(num_speakers=152, speaker_offset=225), "multi-hop": dict(sa_num_hop=2, dec_sa_num_hop=2),
}
shapes = [(32, 160, 800), (32, 160, 500), (32, 97, 330)]
only = [x for x in os.environ.get("SWEEP_ONLY", "").split(",") if x]      # e.g. SWEEP_ONLY="baseline tacotron,default"
for prec in os.environ.get("SWEEP_PREC", "bf16,f32").split(","):
    ops.set_precision(prec)
    for name, kw in variants.items():
        if only and name not in only:
            continue
        for (B, Ti, Tm) in shapes:
            cfg = ModelConfig(**kw)
            extra = dict(num_speakers=152, speaker_offset=225) if cfg.num_speakers else {}
            eng = Engine(cfg, "cuda", param_seed=0, rng_seed=1)
            b = eng.to_device_batch(synthetic_batch(B, Ti, Tm, seed=1234, **extra))
            try:
                for _ in range(2): ctx = eng.train_step(b)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                n = 6 if prec == "bf16" else 3
                for _ in range(n): ctx = eng.train_step(b)
                torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
                eng.check_clusters(ctx)
                ok = torch.isfinite(eng.losses).all().item()
                print("%-4s %-20s B=%d Ti=%d Tm=%d: %8.2f ms/step loss %.4f %s" % (prec, name, B, Ti, Tm, ms, float(eng.losses[2]), "" if ok and ms < 400 else "  <-- CHECK"))
            except Exception as e:
                print("%-4s %-20s B=%d Ti=%d Tm=%d: FAILED %s" % (prec, name, B, Ti, Tm, str(e)[:120]))
            del eng
ops.set_precision("bf16")
