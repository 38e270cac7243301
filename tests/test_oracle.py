"""CPU tests of the oracle itself (-m "not gpu"): the two independent restatements agree, the committed golden
fixtures reproduce, gradients of the torch restatement match finite differences of the NumPy one, and the
reference's only unit test (train == incremental inference for the decoder self-attention,
reference modules/transformer_test.py:40-82) holds as a property of the restated math."""
import os

import numpy as np
import pytest
import torch

from common import MEDIUM, SMALL, make_params, oracle_cfg, oracle_run, small_batch
from oracle import numpy_ref, rng, torch_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_gold(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    P = {k[6:]: z[k] for k in z.files if k.startswith("param.")}
    batch = {k[6:]: z[k] for k in z.files if k.startswith("batch.")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("out.")}
    return P, batch, out, int(z["seed"])


@pytest.mark.parametrize("name,cfg_kw", [("small", SMALL), ("medium", MEDIUM)])
def test_golden_fixtures_reproduce(name, cfg_kw):
    P, batch, gold, seed = load_gold(name)
    out_np = numpy_ref.forward(P, batch, oracle_cfg(cfg_kw), True, seed=seed)
    out_t = torch_ref.forward(torch_ref.to_torch(P), torch_ref.batch_to_torch(batch), oracle_cfg(cfg_kw), True, seed)
    for k in ("mel", "stop", "alignment1", "alignment2", "lstm_out", "sa_out"):
        assert np.abs(out_np[k] - gold[k]).max() < 1e-6, k
        assert np.abs(out_t[k].detach().numpy() - gold[k]).max() < 1e-6, k
    assert abs(out_np["loss"] - float(gold["loss"])) < 1e-12
    assert abs(float(out_t["loss"]) - float(gold["loss"])) < 1e-10


MULTI_HOP = dict(SMALL, sa_num_hop=2, dec_sa_num_hop=3)     # stacked SelfAttentionTransformer blocks (modules/module.py:411-419, :707-715)


@pytest.mark.parametrize("kw", [SMALL, MULTI_HOP])
def test_numpy_and_torch_restatements_agree_eval_mode(kw):
    cfg, P = make_params(kw, seed=5)
    batch = small_batch(cfg, 2, 7, 10, seed=9)
    a = numpy_ref.forward(P, batch, oracle_cfg(kw), False, seed=0)
    # eval mode needs BN moving stats in the torch restatement only for inference; compare training=True instead
    b = torch_ref.forward(torch_ref.to_torch(P), torch_ref.batch_to_torch(batch), oracle_cfg(kw), True, 3)
    c = numpy_ref.forward(P, batch, oracle_cfg(kw), True, seed=3)
    assert abs(float(b["loss"]) - c["loss"]) < 1e-12
    assert np.isfinite(a["loss"])
    if kw is MULTI_HOP:        # the extra hops have their own weights and change the result
        assert "enc.sa.h1.kvq.W" in P and "dec.sa.h2.t.W" in P
        c1 = numpy_ref.forward(P, batch, oracle_cfg(SMALL), True, seed=3)
        assert abs(c1["loss"] - c["loss"]) > 1e-6


def test_torch_gradients_match_numpy_finite_differences():
    cfg, P = make_params(SMALL, seed=2)
    batch = small_batch(cfg, 2, 6, 8, seed=4)
    _, _, g = oracle_run(SMALL, P, batch, True, seed=5)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    rs = np.random.default_rng(0)
    for name in ("dec.att1.U", "dec.att.Wq", "enc.lstm_fw.W", "dec.lstm1.W", "enc.bank2.W", "dec.sa.kvq.W",
                 "dec.att1.F", "enc.proj1.gamma"):
        idx = tuple(rs.integers(0, s) for s in P64[name].shape)
        eps = 1e-5
        Pp = dict(P64); Pp[name] = P64[name].copy(); Pp[name][idx] += eps
        Pm = dict(P64); Pm[name] = P64[name].copy(); Pm[name][idx] -= eps
        fd = (numpy_ref.forward(Pp, batch, oracle_cfg(SMALL), True, 5)["loss"] -
              numpy_ref.forward(Pm, batch, oracle_cfg(SMALL), True, 5)["loss"]) / (2 * eps)
        assert abs(fd - g[name][idx]) < 1e-6 + 1e-4 * abs(fd), (name, fd, g[name][idx])


def test_invariants():
    cfg, P = make_params(MEDIUM, seed=3)
    batch = small_batch(cfg, 3, 19, 22, seed=6)
    out = numpy_ref.forward(P, batch, oracle_cfg(MEDIUM), True, seed=1)
    assert np.allclose(out["alignment1"].sum(-1), 1) and np.allclose(out["alignment2"].sum(-1), 1)
    for b, L in enumerate(batch["source_length"]):
        assert np.all(out["alignment1"][b, :, L:] == 0)
        assert np.all(out["lstm_out"][b, L:] == 0)          # dynamic_rnn zero output beyond sequence_length
    # loss equals the hand-computed masked mean
    w = batch["spec_loss_mask"][:, :, None]
    ml = (np.abs(out["mel"] - batch["mel"]) * w).sum() / (cfg.num_mels * batch["spec_loss_mask"].sum())
    assert abs(ml - out["mel_loss"]) < 1e-12


def test_batch_contract():
    """reference datasets/ljspeech/dataset.py:127-167,264-281: padding values and mask layout."""
    cfg, _ = make_params(SMALL)
    b = small_batch(cfg, 4, 9, 12, seed=1)
    for i in range(4):
        L, n = int(b["source_length"][i]), int(b["target_length"][i])
        assert n % cfg.r == 0
        assert np.all(b["source"][i, L:] == 0) and b["source"][i, 0] == 0 and b["source"][i, L - 1] == 0
        assert np.all(b["mel"][i, n:] == -3.0)
        assert np.all(b["done"][i, :n // cfg.r - 1] == 0) and np.all(b["done"][i, n // cfg.r - 1:] == 1)
        assert np.all(b["spec_loss_mask"][i, :n] == 1) and np.all(b["spec_loss_mask"][i, n:] == 0)
        assert np.all(b["binary_loss_mask"][i, :n // cfg.r] == 1) and np.all(b["binary_loss_mask"][i, n // cfg.r:] == 0)


def test_train_equals_incremental_decoder_self_attention():
    """Property pinned by the reference's only test (modules/transformer_test.py:40-82): the batched causal
    self-attention used in training equals the step-by-step history re-evaluation used at inference
    (TransformerWrapper, modules/rnn_wrappers.py:111-124) — restated on the oracle's math, dropout 0."""
    g = np.random.default_rng(0)
    for trial in range(12):
        B, r = int(g.integers(1, 4)), int(g.integers(1, 3))
        T, D = int(g.integers(2, 9)) * r, int(g.integers(1, 11)) * 2
        x = torch.tensor(g.integers(-1, 2, (B, T, D)).astype(np.float64))
        P = {"p.kvq.W": torch.tensor(g.normal(0, 0.5, (D, 3 * D))), "p.kvq.b": torch.tensor(g.normal(0, 0.1, 3 * D)),
             "p.o.W": torch.tensor(g.normal(0, 0.5, (D, D))), "p.o.b": torch.tensor(g.normal(0, 0.1, D)),
             "p.t.W": torch.tensor(g.normal(0, 0.5, (D, D))), "p.t.b": torch.tensor(g.normal(0, 0.1, D))}
        full, _ = torch_ref.self_attention_transformer(x, P, "p", 2, True, 0.0, False, 0, rng.STREAM_DEC_SA)
        for t in range(T):
            inc, _ = torch_ref.self_attention_transformer(x[:, :t + 1], P, "p", 2, True, 0.0, False, 0, rng.STREAM_DEC_SA)
            assert torch.allclose(inc[:, -1], full[:, t], atol=1e-12)


def test_mask_generator_statistics_and_determinism():
    m = rng.keep_mask(5, 9, (200, 300), 0.5)
    assert abs(m.mean() - 0.5) < 0.01
    assert np.array_equal(m, rng.keep_mask(5, 9, (200, 300), 0.5))
    assert not np.array_equal(m, rng.keep_mask(6, 9, (200, 300), 0.5))
    assert rng.keep_mask(5, 9, (10,), 0.0).all()


def _moving(ocfg, seed=11):
    g = np.random.default_rng(seed)
    dims = {"bank": ocfg.conv_channels * ocfg.max_filter_width, "proj1": ocfg.proj1, "proj2": ocfg.proj2}
    return {n: (torch.as_tensor(g.normal(0, 0.2, d)), torch.as_tensor(g.uniform(0.5, 1.5, d))) for n, d in dims.items()}


@pytest.mark.parametrize("cfg_kw,B,Ti,Tm", [(SMALL, 3, 9, 12), (MEDIUM, 2, 17, 16), (MULTI_HOP, 2, 9, 12)])
def test_whole_decoder_validation_pass_equals_batched_forward(cfg_kw, B, Ti, Tm):
    """the reference's test property (modules/transformer_test.py:40-82) for the WHOLE decoder: the step-by-step
    validation pass (history re-evaluated every step, is_training=False) equals the batched training-branch graph"""
    cfg, P = make_params(cfg_kw, seed=1)
    batch = small_batch(cfg, B, Ti, Tm, seed=3)
    ocfg = oracle_cfg(cfg_kw)
    Pt, bt = torch_ref.to_torch(P), torch_ref.batch_to_torch(batch)
    mv = _moving(ocfg)
    out = torch_ref.infer(Pt, bt["source"], bt["source_length"], ocfg, None, mv, teacher=bt["mel"])
    lo, so, _ = torch_ref.encoder(bt["source"], bt["source_length"], Pt, ocfg, False, 0, bn_moving=mv)
    mel, stop, a1, a2, _ = torch_ref.decoder(lo, so, bt["source_length"], bt["mel"], Pt, ocfg, False, 0)
    for got, want in ((out["mel"], mel), (out["stop"], stop), (out["alignment1"], a1), (out["alignment2"], a2)):
        assert float((got - want).abs().max()) < 1e-12


def test_free_running_decode_feeds_back_and_stops():
    cfg, P = make_params(SMALL, seed=2)
    P = dict(P)
    b = np.array(P["dec.out.b"], dtype=np.float64).copy(); b[-1] = 50.0
    P["dec.out.b"] = b
    batch = small_batch(cfg, 3, 9, 12, seed=5)
    ocfg = oracle_cfg(SMALL)
    bt = torch_ref.batch_to_torch(batch)
    out = torch_ref.infer(torch_ref.to_torch(P), bt["source"], bt["source_length"], ocfg, 30, _moving(ocfg), min_steps=4)
    assert out["steps"] == 6 and out["mel"].shape == (3, 6 * ocfg.r, ocfg.num_mels)
    assert torch.allclose(out["alignment1"].sum(-1), torch.ones(3, 6, dtype=torch.float64))


def test_oracle_forced_alignment_reproduces_free_run():
    """oracle-level property of the forced-alignment mode (modules/teacher_forcing_attention.py:13-78): the mechanisms
    are bypassed and nothing else changes, so feeding a free run's own alignments back reproduces the run."""
    import torch
    from common import SMALL, make_params, small_batch
    from oracle import torch_ref
    cfg, P = make_params(SMALL, seed=2)
    batch = small_batch(cfg, 3, 9, 2 * cfg.r, seed=5)
    ocfg = torch_ref.Cfg(**SMALL)
    Pt = torch_ref.to_torch(P)
    src, sl = torch.as_tensor(batch["source"]), torch.as_tensor(batch["source_length"])
    mv = _moving(ocfg)
    free = torch_ref.infer(Pt, src, sl, ocfg, 7, mv, min_steps=10 ** 6)
    again = torch_ref.infer(Pt, src, sl, ocfg, 7, mv, min_steps=10 ** 6,
                            teacher_alignments=(free["alignment1"], free["alignment2"]))
    assert torch.allclose(free["mel"], again["mel"], atol=1e-12) and torch.allclose(free["stop"], again["stop"], atol=1e-12)


@pytest.mark.parametrize("attention,cumulative", [("location_sensitive", False), ("location_sensitive", True), ("forward", True)])
def test_attention_options_numpy_vs_torch_and_finite_differences(attention, cumulative):
    """hparams attention=location_sensitive / cumulative_weights=True (reference modules/attentions.py:35-42,
    modules/forward_attention.py:118-119): the two independent restatements agree, the options change the result, and
    the autograd gradient of the location filter (the parameter the options act through) matches finite differences."""
    kw = dict(SMALL, attention=attention, cumulative_weights=cumulative)
    cfg, P = make_params(SMALL, seed=5)
    batch = small_batch(cfg, 2, 7, 10, seed=9)
    a = numpy_ref.forward(P, batch, oracle_cfg(kw), True, seed=3)
    b = torch_ref.forward(torch_ref.to_torch(P), torch_ref.batch_to_torch(batch), oracle_cfg(kw), True, 3)
    base = numpy_ref.forward(P, batch, oracle_cfg(SMALL), True, seed=3)
    assert abs(float(b["loss"]) - a["loss"]) < 1e-12
    assert np.abs(b["alignment1"].detach().numpy() - a["alignment1"]).max() < 1e-12
    assert np.abs(a["alignment1"] - base["alignment1"]).max() > 1e-6
    assert np.allclose(a["alignment1"].sum(-1), 1.0)
    _, _, g = oracle_run(kw, P, batch, True, seed=3)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    for name, idx in (("dec.att1.F", (1, 0, 2)), ("dec.att1.U", (0, 3))):
        eps = 1e-5
        Pp = dict(P64); Pp[name] = P64[name].copy(); Pp[name][idx] += eps
        Pm = dict(P64); Pm[name] = P64[name].copy(); Pm[name][idx] -= eps
        fd = (numpy_ref.forward(Pp, batch, oracle_cfg(kw), True, 3)["loss"] -
              numpy_ref.forward(Pm, batch, oracle_cfg(kw), True, 3)["loss"]) / (2 * eps)
        assert abs(fd - g[name][idx]) < 1e-6 * max(1.0, abs(fd)), (name, fd, g[name][idx])
