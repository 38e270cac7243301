#!/usr/bin/env python
"""Hand-off time-outs of the layer pipeline over a grid of shapes (B above the 32-sample envelope x decoder lengths): per shape,
6 unsynchronised train steps, then the sticky error words.  usage: python tools/residency_probe.py [B,Ti,Tm ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satt_amd  # noqa: F401
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
ops.set_precision("bf16")
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or \
    [(B, Ti, Tm) for B in (33, 36, 40, 42) for (Ti, Tm) in ((120, 500), (100, 400), (88, 442), (143, 768), (68, 306), (40, 210))]
shared = os.environ.get("SATT_PROBE_SHARED") == "1"        # one engine for every shape (what a length-bucketed corpus does)
eng0 = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1) if shared else None
for (B, Ti, Tm) in shapes:
    eng = eng0 or Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
    b = eng.to_device_batch(synthetic_batch(B, Ti, Tm, seed=100))
    t0 = time.perf_counter()
    try:
        for _ in range(6):
            if os.environ.get("SATT_PROBE_FWD_ONLY") == "1":
                ctx = eng.forward(b, True)
            else:
                ctx = eng.train_step(b); eng.optimizer_step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 6 * 1e3
        eng.check_clusters(ctx)
        msg = "ok"
    except Exception as e:
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 6 * 1e3
        msg = "TIMEOUT: %s" % str(e)[:80]
        # -DSATT_XCHG_DEBUG builds: the 16 tail words of every cluster workspace (cluster_xchg.h gather_poll: [0] error, [1] / [2] same-XCD /
        # write-through workgroup-launches, [3] reporters, [4] tag waited for, [5] count, [6] block x | y << 16, [7] first missing slot, [8] its
        # tag, [9] XCC, [10] / [11] workgroups started / finished when the first reporter gave up, [12] / [13] started / finished now, [14] ok mask)
        for (kind, key), ws in (eng._ws_cache or {}).items():
            tail = ws[-64:].view(torch.int32).cpu().tolist() if ws.dtype == torch.uint8 else ws.view(torch.uint8)[-64:].view(torch.int32).cpu().tolist()
            if tail[0]:
                print("    %s %s tail: %s" % (kind, key, tail), flush=True)
        try:
            eng.recover_from_handoff_timeout()        # clears the sticky words (and falls back to the chunked schedule)
            eng.single_launch_attention = True
        except Exception:
            pass
    r = eng.residency
    print("B=%d Ti=%d Tm=%d (Td=%d): attention %d + LSTM %d CUs, fits=%s, chunks %s: %.2f ms/step  %s" %
          (B, Ti, Tm, Tm // 2, r["attention_cus"], r["lstm_cus"], r["fits"], eng._chunk_bounds(Tm // 2, eng.pipeline_chunks)[-3:], ms, msg), flush=True)
    if not shared:
        del eng
