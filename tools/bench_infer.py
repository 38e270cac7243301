#!/usr/bin/env python
"""BASELINE.json config 5: free-running decode (B=1, Ti=100, LJSpeech config) - decoder steps per second.
usage: python tools/bench_infer.py [--steps 200] [--batch 1]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import satt_amd  # noqa: F401
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.inference import infer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--steps-per-graph", type=int, default=8)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--mega-max-b", type=int, default=None, help="largest batch that takes the persistent step kernel (default: the session's)")
a = ap.parse_args()
ops.set_precision(a.precision)
if a.mega_max_b is not None:
    from satt_amd.inference import DecodeSession
    DecodeSession.MEGA_MAX_B = a.mega_max_b
cfg = ModelConfig()
eng = Engine(cfg, "cuda", param_seed=0, rng_seed=1)
g = np.random.default_rng(1234)
B, Ti = a.batch, 100
src = g.integers(1, 68, (B, Ti)); src[:, 0] = 0; src[:, -1] = 0
sl = np.full((B,), Ti, dtype=np.int64)
ap_kw = dict(max_steps=a.steps, min_steps=10 ** 6, check_every=a.steps_per_graph, use_graph=not a.no_graph)
infer(eng, src, sl, **ap_kw)            # warm-up: builds the session of this shape (buffers + captured hipGraph)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = infer(eng, src, sl, **ap_kw)      # encoder + memories + every decoder step + result copies
torch.cuda.synchronize()
dt_all = time.perf_counter() - t0
dt = out["decode_ms"] * 1e-3            # the decoder steps alone (HIP events around the replay loop)
al = out["alignment1"]
frames = a.steps * cfg.r * B
print(json.dumps({"metric": "free-running decode (config 5)", "batch": B, "Ti": Ti, "decoder_steps": out["steps"],
                  "ms_per_step": 1e3 * dt / a.steps, "utterance_ms_incl_encoder": 1e3 * dt_all,
                  "steps_per_graph": a.steps_per_graph, "graph": not a.no_graph,
                  "mel_frames_per_sec": frames / dt,
                  "realtime_factor": (dt / B) / (a.steps * cfg.r * 0.0125), "dtype": a.precision,
                  "alignment_rows_sum_to_one": bool(torch.allclose(al.sum(-1), torch.ones_like(al.sum(-1)), atol=1e-4)),
                  "finite": bool(torch.isfinite(out["mel"]).all())}))
