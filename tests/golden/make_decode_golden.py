"""Freezes the float64 oracle's FREE-RUNNING decode (oracle/torch_ref.py:infer) at BASELINE.json config 5's own workload.

The inference branch this restates: reference modules/module.py:762-778 (RNNTransformer else-branch), modules/rnn_wrappers.py
:47-124,188-214 (history / transformer / stop-token wrappers), StopTokenBasedInferenceHelper semantics (feed back the last
n_feed_frame * num_mels outputs; stop when sigmoid(stop) > 0.5 for every sample and t > min_steps).

Cases (production dimensions = ModelConfig() = examples/ljspeech/self-attention-tacotron.json, `init_params(cfg, 0)` - the
weights bench.py's decode leg uses -, BatchNorm on seeded non-trivial moving statistics, zoneout in interpolation mode):

  * `b1`: B=1, Ti=100, the source bench.py:decode_bench draws (rng 1234), STEPS=200 free-running steps (fixed count);
  * `b8`: B=8, Ti=100 with ragged source lengths 57..100, 200 steps;
  * `b2`: B=2, Ti=100, ragged (the two-sample instantiation of the persistent step kernel), 200 steps.

Kept per case: every stop logit, the argmax path of both alignments, the per-step mean |mel| (a cheap drift detector over the
whole feedback chain), NROW sampled (sample, step) rows of mel / both alignments, and - b1 only - the full mel.  The stop RULE
is pinned without a second oracle run: the stop logit does not feed back, so adding a constant to `dec.out.b[-1]` changes
nothing but the decision; `stop_shift` / `stop_steps` hold a shift for which the rule (min_steps=10, threshold 0.5) fires at a
step where the deciding logit rises fastest; `stop_margin` is the distance of every decision up to there from the threshold.

A second pass with every weight matrix rounded to bf16 (what the hipGraph path's weight shadows hold) gives the size of the
error the bf16 path CAN have through the 200-step feedback chain; it is stored as `bf16w_*` so that the GPU test's bars can be
read against it.  "Parity unpinned" with respect to TF stays true.  CPU only, a few minutes:
    python tests/golden/make_decode_golden.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

STEPS = 200
NROW = 64
PARAM_SEED = 0
BN_SEED = 11
# *_sharp (r6): the same decode with the location-sensitive score reshaped so that alignment 1 is near one-hot (make_bench_golden.py:
# sharpen_params; without dropout and with |mel| < 0.2 fed back the query barely moves, so the decode needs larger scales than the
# training fixtures: v x 100 / x 60, query layer and keys x 4 / x 8 - mean row entropy 0.04 / 0.15 nats, max alpha > 0.95 on > 98 % of
# the rows): the forward variable rides on its +1e-7 floor (reference modules/forward_attention.py:107-121) through the 200-step chain
CASES = {"b1": dict(B=1, Ti=100), "b8": dict(B=8, Ti=100), "b2": dict(B=2, Ti=100),
         "b1_sharp": dict(B=1, Ti=100, sharpen=dict(keep=0, sv=100.0, sq=4.0, sk=4.0)),
         "b2_sharp": dict(B=2, Ti=100, sharpen=dict(keep=0, sv=60.0, sq=8.0, sk=8.0))}


def decode_inputs(B, Ti):
    """the source of bench.py:decode_bench for B=1; for B>1 more rows of the same generator and ragged lengths"""
    g = np.random.default_rng(1234)
    src = g.integers(1, 68, (B, Ti))
    sl = np.full((B,), Ti, dtype=np.int64)
    if B > 1:
        sl = np.sort(np.random.default_rng(4321).integers(57, Ti + 1, B))[::-1].copy().astype(np.int64)
        sl[0] = Ti
    for b in range(B):
        src[b, 0] = 0
        src[b, sl[b] - 1] = 0
        src[b, sl[b]:] = 0
    return src.astype(np.int64), sl


def moving_stats(cfg):
    """seeded non-trivial BatchNorm moving statistics: name -> (mean, var) float32 arrays, in the engine's buffer order"""
    g = np.random.default_rng(BN_SEED)
    out = {}
    for n, c in (("bank", cfg.max_filter_width * cfg.conv_channels), ("proj1", cfg.proj1), ("proj2", cfg.proj2)):
        out[n] = (g.normal(0, 0.2, c).astype(np.float32), g.uniform(0.5, 1.5, c).astype(np.float32))
    return out


def stop_rule_step(logits, shift, min_steps=10):
    """first step index t (0-based) with t > min_steps and every sample's shifted logit > 0, +1 = number of steps run"""
    ok = (logits + shift > 0).all(0)
    for t in range(logits.shape[1]):
        if t > min_steps and ok[t]:
            return t + 1
    return logits.shape[1]


def pick_stop_shift(logits, min_steps=10):
    """the bias shift that makes the rule fire at the step where the deciding logit (min over samples) rises fastest: returns
    (shift, steps run, margin) - every decision up to the firing step is `margin` away from the threshold after the shift"""
    worst = logits.min(0)                     # the sample that decides `all`
    best = None
    for target in range(min_steps + 2, logits.shape[1] - 20):
        lo = -worst[target]                               # shift > lo: fires at `target`
        hi = -worst[min_steps + 1:target].max()           # shift < hi: not before
        if best is None or (hi - lo) > best[2] * 2:
            best = (float(0.5 * (lo + hi)), target + 1, float(0.5 * (hi - lo)))
    if best is None or best[2] <= 0:
        raise RuntimeError("no stop shift found")
    return best


def run(name):
    import satt_amd  # noqa: F401
    from satt_amd.params import ModelConfig, init_params
    from oracle import torch_ref
    case = CASES[name]
    B, Ti = case["B"], case["Ti"]
    cfg = ModelConfig()
    P = init_params(cfg, PARAM_SEED)
    if case.get("sharpen"):
        from golden.make_bench_golden import sharpen_params
        P = sharpen_params(P, **case["sharpen"])
    src, sl = decode_inputs(B, Ti)
    mvn = moving_stats(cfg)
    mv = {k: (torch.as_tensor(m, dtype=torch.float64), torch.as_tensor(v, dtype=torch.float64)) for k, (m, v) in mvn.items()}
    ocfg = torch_ref.Cfg()
    t0 = time.time()
    ref = torch_ref.infer(torch_ref.to_torch(P), torch.as_tensor(src), torch.as_tensor(sl), ocfg, STEPS, mv, min_steps=10 ** 6)
    print("%s: float64 decode of %d steps: %.0f s" % (name, STEPS, time.time() - t0), flush=True)
    # the same with bf16-rounded weight matrices (biases, BN, embeddings' consumers keep fp32 as in the engine's shadows)
    Pb = {k: (torch.as_tensor(np.asarray(v, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()
              if (np.asarray(v).ndim >= 2) else v) for k, v in P.items()}
    rb = torch_ref.infer(torch_ref.to_torch(Pb), torch.as_tensor(src), torch.as_tensor(sl), ocfg, STEPS, mv, min_steps=10 ** 6)
    mel = ref["mel"].numpy(); melb = rb["mel"].numpy()
    stop = ref["stop"].numpy()[..., 0]
    al1, al2 = ref["alignment1"].numpy(), ref["alignment2"].numpy()
    r, nm = cfg.r, cfg.num_mels
    step_mel = mel.reshape(B, STEPS, r * nm)
    g = np.random.default_rng(77)
    rb_, rt_ = g.integers(0, B, NROW).astype(np.int32), g.integers(0, STEPS, NROW).astype(np.int32)
    rt_[:4] = 0; rt_[4:8] = STEPS - 1; rt_[8:12] = 1
    shift, nstop, margin = pick_stop_shift(stop)
    assert stop_rule_step(stop, shift) == nstop
    keep = dict(source=src, source_length=sl, stop=stop.astype(np.float32),
                path1=al1.argmax(-1).astype(np.int16), path2=al2.argmax(-1).astype(np.int16),
                step_abs_mel=np.abs(step_mel).mean(-1).astype(np.float32),
                rows_b=rb_, rows_t=rt_, mel_rows=step_mel[rb_, rt_].astype(np.float32),
                align1_rows=al1[rb_, rt_].astype(np.float32), align2_rows=al2[rb_, rt_].astype(np.float32),
                stop_shift=np.float64(shift), stop_steps=np.int64(nstop), stop_margin=np.float64(margin),
                bf16w_mel_abs_err=np.abs(melb - mel).reshape(B, STEPS, -1).max(-1).astype(np.float32),
                bf16w_stop_abs_err=np.abs(rb["stop"].numpy()[..., 0] - stop).astype(np.float32),
                bf16w_path1_agree=np.float64((rb["alignment1"].numpy().argmax(-1) == al1.argmax(-1)).mean()),
                mel_abs_max=np.float64(np.abs(mel).max()),
                align1_mean_entropy=np.float64(-(al1 * np.log(np.maximum(al1, 1e-300))).sum(-1).mean()),
                align1_frac_max_above_095=np.float64((al1.max(-1) > 0.95).mean()))
    print("%s: alignment-1 mean row entropy %.3f nats, rows with max > 0.95: %.3f" % (name, keep["align1_mean_entropy"], keep["align1_frac_max_above_095"]), flush=True)
    if B == 1:
        keep["mel"] = mel.astype(np.float32)
    for k, (m, v) in mvn.items():
        keep["bn_mean." + k] = m
        keep["bn_var." + k] = v
    path = os.path.join(HERE, "decode_ljspeech_%s.npz" % name)
    np.savez_compressed(path, steps=np.int64(STEPS), param_seed=np.int64(PARAM_SEED), **keep)
    e = keep["bf16w_mel_abs_err"]
    print("%s: wrote %s (%.0f KB); |mel| max %.3f; bf16-weight oracle vs float64: mel abs err max %.3e (step 10: %.2e, 100: %.2e, "
          "199: %.2e), stop %.3e, path1 agreement %.4f; stop rule: shift %.4f -> %d steps (margin %.2e)"
          % (name, path, os.path.getsize(path) / 1024, keep["mel_abs_max"], e.max(), e[:, 10].max(), e[:, 100].max(),
             e[:, 199].max(), keep["bf16w_stop_abs_err"].max(), keep["bf16w_path1_agree"], shift, nstop, margin), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("SATT_ORACLE_THREADS", "6")))
    for n in (sys.argv[1:] or list(CASES)):
        run(n)
