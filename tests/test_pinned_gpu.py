"""GPU tests that do NOT call the live oracle for their expected values, and tests at the exact benchmark workloads.

* `test_hip_matches_golden_fixtures`: the HIP path against the COMMITTED vectors of tests/golden/*.npz (made by
  tests/golden/make_golden.py from oracle/numpy_ref.py and re-checked on CPU by tests/test_oracle.py).  Every other GPU
  parity test recomputes its expectation with the live oracle, so an oracle + kernel co-drift would stay green there;
  here the expectation is frozen data.
* `test_bench_workload_b32_full_length`: BASELINE configs[1] itself - B=32, Ti=160, Tm=800, bf16, the single-launch
  attention schedule, same-XCD exchange - the 128-workgroup launches with their in-kernel chunk hand-offs that only
  bench.py used to see.  The float64 oracle needs minutes per sample at this size, so the judge here is (i) the
  invariants the domain offers and (ii) the exact-fp32-GEMM mode of the same engine on the same batch and masks.
* `test_vctk_workload_b32`: BASELINE configs[3] at its own shape (Ti <= 80, Tm <= 500, 152 speakers).
* `test_steady_state_steps_have_no_handoff_stalls`: back-to-back train steps (no host synchronisation inside) at B=32 over a
  spread of decoder lengths and batch sizes (B > 32: the cluster kernels of two layers no longer fit side by side - the engine
  must not pipeline them).  Found in round 3: with the two decoder LSTM layers on two streams, Td = 250 (any Ti) left two LSTM
  cluster launches partially resident at the start of the backward loop - 1.2 s hand-off timeouts, several per step, invisible to
  the single-shot parity tests (a host synchronisation between forward and backward hides it) and to the Td = 400 benchmark.
"""
import os

import numpy as np
import pytest
import torch

from common import MEDIUM, SMALL

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_gold(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    P = {k[6:]: z[k] for k in z.files if k.startswith("param.")}
    batch = {k[6:]: z[k] for k in z.files if k.startswith("batch.")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("out.")}
    return P, batch, out, int(z["seed"])


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("clusters", [True, False])
@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("name,cfg_kw", [("small", SMALL), ("medium", MEDIUM)])
def test_hip_matches_golden_fixtures(name, cfg_kw, prec, clusters):
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    P, batch, gold, seed = load_gold(name)
    ops.set_precision(prec)
    try:
        eng = Engine(ModelConfig(**cfg_kw), "cuda", params=P, rng_seed=seed)
        eng.use_clusters = clusters
        ctx = eng.forward(eng.to_device_batch(batch), training=True)
        torch.cuda.synchronize()
        eng.check_clusters(ctx)
        out = {k: v.detach().float().cpu().numpy() for k, v in eng.outputs(ctx).items()}
    finally:
        ops.set_precision("bf16")
    keys = ("mel", "stop", "alignment1", "alignment2", "lstm_out", "sa_out", "dec_out")
    errs = {k: rel(out[k], gold[k]) for k in keys}
    errs.update({k: abs(float(out[k]) - float(gold[k])) for k in ("loss", "mel_loss", "done_loss")})
    print(name, prec, {k: "%.2e" % e for k, e in errs.items()})
    if prec == "f32":                                     # exact-fp32 GEMM mode: fp32 tolerance against the frozen float64 vectors
        bad = {k: e for k, e in errs.items() if not e < 2e-4}
    else:                                                 # benchmark precision: BASELINE.json's mel-L1 bar + loose element bounds
        bad = {k: e for k, e in errs.items() if not e < (1e-3 if k.endswith("loss") else 5e-2)}
    assert not bad, bad


def _run_full(cfg, batch, prec, param_seed=3, rng_seed=5, single=True, params=None):
    from satt_amd import ops
    from satt_amd.engine import Engine
    ops.set_precision(prec)
    try:
        eng = Engine(cfg, "cuda", rng_seed=rng_seed, **(dict(params=params) if params is not None else dict(param_seed=param_seed)))
        eng.single_launch_attention = single
        b = eng.to_device_batch(batch)
        for _ in range(2):                 # the second pass runs on recycled buffers (stale but plausible contents)
            eng.zero_grad()
            ctx = eng.forward(b, True)
            eng.backward(ctx)
            torch.cuda.synchronize()
            eng.check_clusters(ctx)
        o = eng.outputs(ctx)
        res = dict(loss=float(o["loss"]), mel_loss=float(o["mel_loss"]), al1=o["alignment1"].cpu().numpy(),
                   al2=o["alignment2"].cpu().numpy(), lstm_out=o["lstm_out"].cpu().numpy(), mel=o["mel"].cpu().numpy(),
                   stop=o["stop"].cpu().numpy(), dec_out=o["dec_out"].float().cpu().numpy(), sa_out=o["sa_out"].float().cpu().numpy(),
                   grad=eng.grad.detach().cpu().numpy().astype(np.float64), G={k: v.detach().cpu().numpy().astype(np.float64)
                                                                               for k, v in eng.G.items()},
                   single=(ctx.get("single_launch_fwd"), ctx.get("single_launch_bwd")))
        return eng, ctx, res
    finally:
        ops.set_precision("bf16")


def _invariants(res, batch):
    al1, al2 = res["al1"], res["al2"]
    assert np.allclose(al1.sum(-1), 1.0, atol=1e-4) and np.allclose(al2.sum(-1), 1.0, atol=1e-4)
    assert (al1 >= 0).all() and (al2 >= 0).all()
    for i, L in enumerate(batch["source_length"]):
        assert np.all(al1[i, :, L:] == 0) and np.all(al2[i, :, L:] == 0)
        assert np.all(res["lstm_out"][i, L:] == 0)
    assert np.isfinite(res["loss"]) and np.isfinite(res["grad"]).all() and np.abs(res["grad"]).max() > 0


def _compare_modes(rb, rf, tag):
    """benchmark precision against the exact-fp32-GEMM mode of the same engine (same batch, same masks)"""
    d_mel, d_loss = abs(rb["mel_loss"] - rf["mel_loss"]), abs(rb["loss"] - rf["loss"])
    e1, e2 = float(np.abs(rb["al1"] - rf["al1"]).max()), float(np.abs(rb["al2"] - rf["al2"]).max())
    a, b = rb["grad"], rf["grad"]
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    worst, wname = 1.0, None
    for k in rb["G"]:
        x, y = rb["G"][k].ravel(), rf["G"][k].ravel()
        c = float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y) + 1e-30))
        if c < worst:
            worst, wname = c, k
    print("[%s] bf16 vs f32 mode: |d mel_loss|=%.3e |d loss|=%.3e max|d align1|=%.3e max|d align2|=%.3e grad cos=%.6f "
          "worst tensor %s cos=%.5f" % (tag, d_mel, d_loss, e1, e2, cos, wname, worst))
    assert d_mel < 1e-3 and d_loss < 2e-3, (d_mel, d_loss)
    # measured (r3, MI355X): B=32 LJSpeech |d mel_loss| 4.2e-6, align1 8.3e-3 (one sharp alignment step shifting by a frame's
    # worth of mass), cos 0.999991, worst tensor 0.9988; VCTK 3.9e-6, 5.8e-3, 0.999987, 0.9982.  Bars = about 3x those distances
    assert e1 < 2.5e-2 and e2 < 1e-3, (e1, e2)
    assert cos > 0.9999 and worst > 0.995, (cos, wname, worst)


# bars of the frozen-oracle tests: (mel-L1 loss, loss, per-sample mel-L1, alignment rows 1 / 2, argmax-path agreement, sampled output
# rows (relative), flat-gradient relative distance by count sketch, worst per-tensor relative distance).  f32 = exact-fp32 GEMM mode
# (only the recurrent weights are consumed as bf16); bf16 = benchmark precision.  BASELINE.json's bar is the first number of
# the bf16 row: |mel L1 - reference| < 1e-3.  The others are about 3x the distances measured on MI355X (profiles/r04_parity_frozen_oracle.log)
FROZEN_BARS = {"f32": dict(mel_loss=1e-5, loss=2e-4, per_sample=2e-5, al1=1.5e-3, al2=5e-5, path=0.999, rows=5e-3, grad=5e-3, tensor=1e-2),
               "bf16": dict(mel_loss=2e-5, loss=5e-4, per_sample=4e-5, al1=1.2e-2, al2=4e-4, path=0.995, rows=3e-2, grad=1.5e-2, tensor=1.5e-1)}
# measured (r4, MI355X; f32 / bf16): LJSpeech mel_loss 1.1e-6 / 5.5e-6, loss 4.0e-5 / 1.3e-4, per-sample 2.8e-6 / 1.3e-5, align1 3.3e-4 / 4.0e-3,
# align2 5.0e-6 / 5.5e-5, path 1.000 / 0.999, rows 1.6e-3 / 9.1e-3, flat gradient 1.6e-3 / 4.7e-3, worst tensor 3.6e-3 (dec.att1.F) / 5.0e-2
# (enc.prenet0.W); VCTK 2.1e-6 / 6.0e-6, 6.0e-5 / 1.6e-4, 4.1e-6 / 1.2e-5, 4.3e-4 / 2.3e-3, 1.4e-5 / 1.2e-4, 1.000 / 0.9997, 1.7e-3 / 7.8e-3,
# 1.3e-3 / 5.6e-3, 2.7e-3 / 5.9e-2.  (The f32 mode's 1.5e-3 gradient distance is the bf16 rounding of the recurrent weights, which
# the persistent kernels consume as bf16 in both modes; weights here come unrounded from init_params.)


# The converged-regime fixtures (r6, VERDICT r5 item 5: tests/golden/make_bench_golden.py `sharpen`): the same B=32 x 400-step workload with
# alignment 1 near one-hot (hi: sum|v| = 443 - the in-chain softmax form; lo: sum|v| = 29.5 - the constant shift / lazy normalisation
# of csrc/attn_cluster.hip).  A near one-hot row moves by a whole position where two neighbouring energies are close, so the
# row / path distances of this regime are larger than the diffuse one's; the loss bars stay BASELINE's.  Bars ~3x the distances
# measured on MI355X (profiles/r06_parity_frozen_oracle.log).
FROZEN_BARS_SHARP = {
    "ljspeech_sharp_lo": {"f32": dict(mel_loss=1e-5, loss=2e-4, per_sample=4e-5, al1=9e-2, al2=6e-5, path=0.99, rows=4e-2, grad=6e-2, tensor=0.3),
                          "bf16": dict(mel_loss=2e-5, loss=5e-4, per_sample=2e-4, al1=0.2, al2=2e-4, path=0.95, rows=0.22, grad=0.16, tensor=0.75)},
    "ljspeech_sharp_hi": {"f32": dict(mel_loss=1e-5, loss=2e-4, per_sample=2e-5, al1=0.1, al2=4e-5, path=0.99, rows=2.5e-2, grad=3e-2, tensor=0.25),
                          "bf16": dict(mel_loss=3e-5, loss=5e-4, per_sample=2.5e-4, al1=0.9, al2=2e-4, path=0.95, rows=0.2, grad=0.2, tensor=3.0)},
}
# measured (r6, MI355X; f32 / bf16): sharp_lo mel_loss 1.6e-6 / 2.5e-6, loss 5.5e-5 / 2.6e-5, per-sample 1.0e-5 / 6.5e-5, align1 rows 2.7e-2 / 6.6e-2,
# align2 1.8e-5 / 6.1e-5, path 0.998 / 0.981, rows 1.4e-2 / 7.3e-2, flat gradient 2.0e-2 / 5.1e-2, worst tensor 9.3e-2 / 0.24 (dec.prenet0.W);
# sharp_hi 1.1e-6 / 7.9e-6, 3.1e-5 / 1.5e-4, 6.1e-6 / 7.5e-5, 3.4e-2 / 0.30 (one sampled one-hot row a position off), 1.0e-5 / 5.9e-5, 0.998 / 0.987,
# 7.5e-3 / 6.4e-2, 9.6e-3 / 6.2e-2, 7.5e-2 (dec.att1.bF) / 0.95 (dec.att1.U: the location term's gradient at sum|v| = 443).  The mel-L1 loss -
# BASELINE's bar, 1e-3 - stays at the diffuse regime's 1e-6 .. 8e-6 in both forms of the forward variable's softmax.


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["ljspeech", "vctk", "ljspeech_sharp_lo", "ljspeech_sharp_hi"])
def test_bench_workload_vs_frozen_float64_oracle(name, prec):
    """BASELINE configs[1] (LJSpeech B=32, Ti=160, Tm=800) and configs[3] (VCTK B=32, Ti=80, Tm=500) - the
    exact batches bench.py times, on the schedule it times (ONE attention launch per direction, same-XCD exchange) - judged
    by the float64 oracle, frozen by tests/golden/make_bench_golden.py into tests/golden/bench_<name>.npz: losses, per-sample
    mel-L1, both alignments (argmax paths + sampled rows), sampled output rows, and the gradient through count sketches
    (flat + per tensor; small tensors in full).  Until r4 these two workloads were only judged by the engine's own f32 mode."""
    from test_model_gpu import assert_same_xcd_fast_path
    from common import count_sketch
    from golden.make_bench_golden import CASES, make_batch, sample_rows, sharpen_params
    from satt_amd.params import ModelConfig, init_params
    z = np.load(os.path.join(GOLD, "bench_%s.npz" % name))
    case = CASES[name]
    cfg = ModelConfig(**case["cfg"])
    batch = make_batch(case["batch"])
    B, Td = batch["done"].shape
    params = None
    if "sharpen" in case:
        params = sharpen_params(init_params(cfg, int(z["meta.param_seed"])), **case["sharpen"])
        print("[frozen oracle] %s: alignment-1 mean row entropy %.3f nats, mean max %.3f, rows with max > 0.95: %.3f, sum|v| = %.1f"
              % (name, float(z["align1_mean_entropy"]), float(z["align1_max_mean"]), float(z["align1_frac_max_above_095"]), float(z["att1_v_abs_sum"])))
    eng, ctx, r = _run_full(cfg, batch, prec, param_seed=int(z["meta.param_seed"]), rng_seed=int(z["meta.rng_seed"]), params=params)
    assert r["single"] == (True, True), r["single"]
    eng.last_ctx = ctx
    assert_same_xcd_fast_path(eng, B)
    _invariants(r, batch)
    bars = FROZEN_BARS_SHARP[name][prec] if name in FROZEN_BARS_SHARP else FROZEN_BARS[prec]
    got = {}
    got["mel_loss"] = abs(r["mel_loss"] - float(z["mel_loss"]))
    got["loss"] = abs(r["loss"] - float(z["loss"]))
    w = batch["spec_loss_mask"].astype(np.float64)
    per = (np.abs(r["mel"].astype(np.float64) - batch["mel"]).mean(-1) * w).sum(-1) / np.maximum(w.sum(-1), 1.0)
    got["per_sample"] = float(np.abs(per - z["per_sample_mel_l1"]).max())
    sb, st = z["rows_b"], z["rows_t"]
    sb2, st2 = sample_rows(B, Td, 99)
    assert np.array_equal(sb, sb2) and np.array_equal(st, st2)
    got["al1"] = float(np.abs(r["al1"][sb, st] - z["align1_rows"]).max())
    got["al2"] = float(np.abs(r["al2"][sb, st] - z["align2_rows"]).max())
    # argmax paths: compared where the oracle's maximum is not a near-tie (two memory rows within 2 % of each other flip freely)
    p1 = r["al1"].argmax(-1)
    top2 = np.sort(r["al1"], -1)[..., -2:]
    clear = top2[..., 1] > 1.02 * top2[..., 0]
    got["path"] = float((p1[clear] == z["path1"][clear]).mean())
    off = np.abs(p1.astype(np.int64) - z["path1"]).max()
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-12))
    rows = dict(mel=rel(r["mel"].reshape(B, Td, -1)[sb, st], z["mel_rows"]), stop=rel(r["stop"].reshape(B, Td)[sb, st], z["stop_rows"]),
                dec_out=rel(r["dec_out"][sb, st], z["dec_out_rows"]), lstm_out=rel(r["lstm_out"][z["enc_b"], z["enc_t"]], z["lstm_out_rows"]),
                sa_out=rel(r["sa_out"][z["enc_b"], z["enc_t"]], z["sa_out_rows"]))
    got["rows"] = max(rows.values())
    names = [str(n) for n in z["grad_names"]]
    flat = np.concatenate([r["G"][k].ravel() for k in names])
    ref_sk = z["grad_sketch_all"]
    got["grad"] = float(np.linalg.norm(count_sketch(flat, int(z["meta.sketch_g"]), 0) - ref_sk) / np.linalg.norm(ref_sk))
    gn = float(np.linalg.norm(flat) / float(z["grad_norm_all"]))
    worst, wname = 0.0, None
    for i, k in enumerate(names):
        nrm = float(z["grad_norms"][i])
        if nrm < 1e-6 * float(z["grad_norm_all"]):
            continue                                  # (a tensor without gradient signal: nothing to be relative to)
        if ("grad_full." + k) in z.files:
            d = float(np.linalg.norm(r["G"][k] - z["grad_full." + k]) / nrm)
        else:
            d = float(np.linalg.norm(count_sketch(r["G"][k], int(z["meta.sketch_t"]), i + 1) - z["grad_sketch." + k]) / nrm)
        if d > worst:
            worst, wname = d, k
    got["tensor"] = worst
    print("[frozen oracle] %s %s: " % (name, prec) + " ".join("%s=%.3e" % kv for kv in got.items()) +
          " | rows %s | max path offset %d | |g|/|g_ref|=%.5f | worst tensor %s" % ({k: "%.2e" % v for k, v in rows.items()}, off, gn, wname))
    bad = {k: v for k, v in got.items() if (v < bars[k] if k == "path" else not v < bars[k])}
    assert not bad, (bad, bars)
    assert abs(gn - 1.0) < (2e-3 if prec == "f32" else 2e-2), gn


def test_bench_workload_b32_full_length():
    from test_model_gpu import assert_same_xcd_fast_path
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    B = 32
    batch = synthetic_batch(B, 160, 800, seed=1234)         # bench.py's own batch (rank 0)
    res = {}
    for prec in ("bf16", "f32"):
        eng, ctx, r = _run_full(ModelConfig(), batch, prec)
        # the schedule bench.py times: ONE attention launch per direction with in-kernel chunk hand-offs, plain-store exchange
        assert r["single"] == (True, True), r["single"]
        eng.last_ctx = ctx
        assert_same_xcd_fast_path(eng, B)
        _invariants(r, batch)
        res[prec] = r
    _compare_modes(res["bf16"], res["f32"], "B=32 Ti=160 Tm=800")
    # the chunked schedule (one attention launch per pipeline chunk: no in-kernel hand-off) must give the same step
    _, _, rc = _run_full(ModelConfig(), batch, "bf16", single=False)
    assert rc["single"] == (False, False)
    assert abs(rc["loss"] - res["bf16"]["loss"]) < 1e-5
    gd = float(np.abs(rc["grad"] - res["bf16"]["grad"]).max())
    assert gd < 1e-3 * float(np.abs(rc["grad"]).max()), gd


def test_vctk_workload_b32():
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    B = 32
    cfg = ModelConfig(num_speakers=152, speaker_offset=225)
    batch = synthetic_batch(B, 80, 500, seed=4321, min_source_length=30, min_target_steps=90, num_speakers=152,
                            speaker_offset=225)
    res = {}
    for prec in ("bf16", "f32"):
        eng, ctx, r = _run_full(cfg, batch, prec)
        _invariants(r, batch)
        assert np.abs(r["G"]["speaker_embedding"]).max() > 0
        used = np.unique(batch["speaker_id"] - 225)
        unused = np.setdiff1d(np.arange(152), used)
        assert np.all(r["G"]["speaker_embedding"][unused] == 0)        # only the batch's speakers receive a gradient
        res[prec] = r
    _compare_modes(res["bf16"], res["f32"], "VCTK B=32 Ti=80 Tm=500")


@pytest.mark.parametrize("B,Ti,Tm", [(32, 80, 500), (32, 160, 500), (32, 160, 800), (32, 120, 640), (32, 100, 1000), (32, 60, 200),
                                      (32, 33, 74), (40, 160, 400), (48, 120, 600), (64, 160, 400), (8, 160, 800), (1, 100, 400)])
def test_steady_state_steps_have_no_handoff_stalls(B, Ti, Tm):
    import time
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
    b = eng.to_device_batch(synthetic_batch(B, Ti, Tm, seed=1234))
    for _ in range(3):
        ctx = eng.train_step(b)
    torch.cuda.synchronize()
    eng.check_clusters(ctx)
    n = 12
    t0 = time.perf_counter()
    for _ in range(n):
        ctx = eng.train_step(b)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    eng.check_clusters(ctx)                     # sticky: a hand-off timeout in ANY of the steps raises here
    assert bool(ctx.get("single_launch_fwd") and ctx.get("single_launch_bwd")) == eng.residency["fits"] == (B <= 32)     # larger batches: layers one after the other
    print("B=%d Ti=%d Tm=%d: %.2f ms/step" % (B, Ti, Tm, ms))
    assert ms < 30.0, ms                        # (a single timeout costs > 1 s)
    assert np.isfinite(float(eng.losses[2]))


@pytest.mark.parametrize("streams", [1, 2])
@pytest.mark.parametrize("B,Ti,Tm", [(32, 100, 400), (33, 100, 400), (36, 80, 384), (40, 100, 400), (48, 60, 400), (64, 100, 400)])
def test_layer_pipeline_is_sized_from_the_resident_capacity(B, Ti, Tm, streams):
    """forward progress of the spinning cluster kernels must not rest on an assumed device.  The
    launchers refuse a grid above the kernel's resident capacity (occupancy calculator x CU count) and the engine keeps the
    attention kernel and the LSTM cluster launches in flight together only if their footprints fit side by side
    (Engine._layers_fit_side_by_side).  Swept here: batch sizes around and above the 32-sample envelope, one and two LSTM streams,
    Td = 200 (tail chunks of 8 steps): back-to-back steps, zero error words, and the schedule the rule says."""
    import time
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
    eng.lstm_one_stream = streams == 1
    b = eng.to_device_batch(synthetic_batch(B, Ti, Tm, seed=77))
    for _ in range(2):
        ctx = eng.train_step(b)
    torch.cuda.synchronize()
    eng.check_clusters(ctx)
    r = eng.residency
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert r["cus"] == cus and r["attention"][0][1] >= 1 and r["lstm"][0][1] >= 1
    assert r["attention"][0][0] == r["attention"][1][0] == B * ctx["att_cluster"][0]
    want = r["attention_cus"] + r["lstm_cus"] <= cus
    assert r["fits"] == want and bool(ctx.get("single_launch_fwd") and ctx.get("single_launch_bwd")) == want, (r, want)
    if B == 32:
        assert want == (streams == 1)           # the benchmark envelope pipelines with ONE LSTM stream; two would not fit
    t0 = time.perf_counter()
    for _ in range(6):
        ctx = eng.train_step(b)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    eng.check_clusters(ctx)                     # sticky: a hand-off timeout in ANY of the steps raises here
    print("B=%d Ti=%d Tm=%d, %d LSTM stream(s): attention %d CUs + LSTM %d CUs of %d -> %s, %.2f ms/step"
          % (B, Ti, Tm, streams, r["attention_cus"], r["lstm_cus"], cus, "side by side" if want else "one after the other", ms))
    assert ms < 30.0 and np.isfinite(float(eng.losses[2]))


def test_cluster_launchers_refuse_a_grid_above_the_resident_capacity():
    """the C-ABI itself: satt_lstm_cluster_fwd with more workgroups than the device can hold returns SATT_E_UNSUPPORTED instead of
    launching members that would spin for peers which cannot become resident"""
    from satt_amd import ops, _lib
    n, per, cus = ops.lstm_cluster_residency(8, 4, 256, 4, False)
    assert n == 32 and per >= 1 and cus == torch.cuda.get_device_properties(0).multi_processor_count
    Bbig = per * cus // 4 + 1                    # one sample too many
    assert ops.lstm_cluster_residency(Bbig, 4, 256, 4, False)[0] > per * cus
    assert ops.lstm_cluster_size(Bbig, 256) in (0, 2)       # the selection asks the same question (a smaller cluster may still fit)
    D = 256
    xg = torch.zeros(1, Bbig * 4, 4 * D, device="cuda")
    W = torch.zeros(D, 4 * D, device="cuda")
    pf, pb = ops.lstm_cluster_pack(W, D, 4)
    e = lambda *s: torch.empty(*s, device="cuda")
    seed = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = ops.lstm_cluster_ws(Bbig, D, 4, "cuda")
    with pytest.raises(_lib.SattError):
        ops.lstm_cluster_fwd(xg, pf, Bbig, 4, D, 4, True, 0.1, 0.1, seed, 12, 13, e(Bbig * 4, D), e(1, Bbig * 4, 4 * D),
                             e(1, Bbig * 4, D), e(1, Bbig * 4, D), e(1, Bbig * 4, D), ws, 0, 4)


@pytest.mark.parametrize("model", ["self-attention", "baseline"])
def test_shapes_changing_from_step_to_step_keep_the_steady_state(model):
    """what a length-bucketed corpus feeds: (B, Ti, Tm) differs from one unsynchronised step to the next (workspaces, packs and
    pipeline chunking switch per step).  No hand-off timeout, and the mixed sequence costs no more than its steps cost alone."""
    import time
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    ops.set_precision("bf16")
    kw = dict() if model == "self-attention" else dict(sa_units=0, att2_units=0, dec_sa_units=0, att1_units=256)
    eng = Engine(ModelConfig(**kw), "cuda", param_seed=0, rng_seed=1)
    shapes = [(32, 143, 768), (24, 88, 442), (32, 68, 306), (8, 25, 178), (40, 120, 500), (16, 133, 632), (32, 40, 210)]
    batches = [eng.to_device_batch(synthetic_batch(B, Ti, Tm, seed=100 + i)) for i, (B, Ti, Tm) in enumerate(shapes)]
    per = []
    for b in batches:
        for _ in range(3):
            ctx = eng.train_step(b); eng.optimizer_step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            ctx = eng.train_step(b); eng.optimizer_step()
        torch.cuda.synchronize(); per.append((time.perf_counter() - t0) / 3)
        eng.check_clusters(ctx)
    order = [int(i) for i in np.random.default_rng(3).integers(0, len(batches), 42)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in order:
        ctx = eng.train_step(batches[i]); eng.optimizer_step()
    torch.cuda.synchronize(); mixed = time.perf_counter() - t0
    eng.check_clusters(ctx)
    expect = sum(per[i] for i in order)
    print("%s: 42 steps in mixed shape order %.1f ms, alone %.1f ms" % (model, mixed * 1e3, expect * 1e3))
    assert mixed < 1.25 * expect + 0.02, (mixed, expect)
    assert np.isfinite(float(eng.losses[2]))
