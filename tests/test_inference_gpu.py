"""GPU parity of the step-by-step decode (inference branch, BASELINE.json config 5 / SURVEY.md §8 a18):
KV-cached incremental causal self-attention + the training kernels restarted per step.
(1) the reference's own test property (modules/transformer_test.py:40-82): the validation pass (teacher-fed,
    step by step) equals the batched teacher-forced forward with is_training=False;
(2) free-running decode against the float64 oracle (oracle/torch_ref.py:infer)."""
import numpy as np
import pytest
import torch

from common import MEDIUM, SMALL, make_params, rel_err, small_batch

pytestmark = pytest.mark.gpu


def make_engine(cfg, P, prec="f32"):
    from satt_amd import ops
    from satt_amd.engine import Engine
    ops.set_precision(prec)
    eng = Engine(cfg, "cuda", params=P, rng_seed=7)
    g = np.random.default_rng(11)
    mv = {}
    for name, (mean, var) in eng.bn.items():        # non-trivial moving statistics, shared with the oracle
        m = g.normal(0, 0.2, mean.shape[0]).astype(np.float32); v = g.uniform(0.5, 1.5, var.shape[0]).astype(np.float32)
        mean.copy_(torch.as_tensor(m)); var.copy_(torch.as_tensor(v))
        mv[name] = (torch.as_tensor(m, dtype=torch.float64), torch.as_tensor(v, dtype=torch.float64))
    return eng, mv


@pytest.mark.parametrize("cfg_kw,B,Ti,Tm", [(MEDIUM, 5, 37, 46), (SMALL, 3, 9, 12), (MEDIUM, 11, 29, 20), (dict(), 2, 41, 24)])
def test_validation_pass_equals_batched_forward(cfg_kw, B, Ti, Tm):
    from satt_amd.inference import infer
    cfg, P = make_params(cfg_kw, seed=1)
    batch = small_batch(cfg, B, Ti, Tm, seed=3)
    eng, _ = make_engine(cfg, P)
    b = eng.to_device_batch(batch)
    ctx = eng.forward(b, training=False)
    ref = {k: v.detach().float().cpu().numpy() for k, v in eng.outputs(ctx).items()}
    out = infer(eng, b["source"], b["source_length"], teacher=b["mel"])
    for k in ("mel", "stop", "alignment1", "alignment2"):
        e = rel_err(out[k].detach().cpu().numpy(), ref[k])
        print(k, e)
        assert e < 2e-5, (k, e)


@pytest.mark.parametrize("cfg_kw,B,Ti,steps", [(MEDIUM, 4, 33, 14), (SMALL, 3, 9, 12)])
def test_free_running_decode_matches_oracle(cfg_kw, B, Ti, steps):
    from oracle import torch_ref
    from satt_amd.inference import infer
    cfg, P = make_params(cfg_kw, seed=2)
    batch = small_batch(cfg, B, Ti, 2 * cfg.r, seed=5)
    eng, mv = make_engine(cfg, P)
    ocfg = torch_ref.Cfg(**cfg_kw)
    Pt = torch_ref.to_torch(P)
    src, sl = torch.as_tensor(batch["source"]), torch.as_tensor(batch["source_length"])
    ref = torch_ref.infer(Pt, src, sl, ocfg, steps, mv, min_steps=10 ** 6)        # fixed number of steps
    out = infer(eng, src, sl, max_steps=steps, min_steps=10 ** 6)
    assert out["steps"] == ref["steps"] == steps
    for k in ("mel", "stop", "alignment1", "alignment2"):
        e = rel_err(out[k].detach().cpu().numpy(), ref[k].numpy())
        print(k, e)
        assert e < 5e-4, (k, e)


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("cfg_kw,B,Ti,steps", [(MEDIUM, 4, 33, 14), (SMALL, 3, 9, 12),
                                               (dict(MEDIUM, num_speakers=7, speaker_dim=16, speaker_offset=225), 3, 17, 9)])
def test_dropout_on_inference(cfg_kw, B, Ti, steps, fuse):
    """apply_dropout_on_inference (reference modules/module.py:564-577): the plain decoder PreNet layers keep their dropout while
    synthesising and in the evaluation pass (MultiSpeakerPreNet does not: :569-570).  The decode kernels draw the stateless
    mask of row (b, step) of a [B, steps, units] activation, so (i) the free run equals the float64 oracle's, (ii) the
    teacher-fed step-by-step pass equals the batched evaluation forward, and (iii) both differ from the dropout-free run."""
    from oracle import torch_ref
    from satt_amd.inference import DecodeSession, infer
    kw = dict(cfg_kw, apply_dropout_on_inference=True)
    cfg, P = make_params(kw, seed=2)
    Tm = steps * cfg.r
    batch = small_batch(cfg, B, Ti, Tm, seed=5)
    if cfg.num_speakers:
        batch["speaker_id"] = (np.random.default_rng(1).integers(0, cfg.num_speakers, B) + cfg.speaker_offset).astype(np.int64)
    eng, mv = make_engine(cfg, P)                                   # engine seed word = 7
    ocfg = torch_ref.Cfg(**kw)
    Pt = torch_ref.to_torch(P)
    bt = torch_ref.batch_to_torch(batch)
    spk = dict(speaker_id=bt["speaker_id"]) if cfg.num_speakers else {}
    espk = dict(speaker_id=batch["speaker_id"]) if cfg.num_speakers else {}
    old = DecodeSession.FUSE
    DecodeSession.FUSE = fuse
    try:
        ref = torch_ref.infer(Pt, bt["source"], bt["source_length"], ocfg, steps, mv, min_steps=10 ** 6, seed=7, **spk)
        out = infer(eng, bt["source"], bt["source_length"], max_steps=steps, min_steps=10 ** 6, dropout_seed=7, **espk)
        for k in ("mel", "stop", "alignment1"):
            e = rel_err(out[k].detach().cpu().numpy(), ref[k].numpy())
            print("free run", k, e)
            assert e < 5e-4, (k, e)
        # (ii) teacher-fed: oracle, and the batched evaluation pass of the engine
        reft = torch_ref.infer(Pt, bt["source"], bt["source_length"], ocfg, None, mv, teacher=bt["mel"], seed=7, **spk)
        outt = infer(eng, bt["source"], bt["source_length"], teacher=bt["mel"].float(), dropout_seed=7, **espk)
        b = eng.to_device_batch(batch)
        fw = eng.outputs(eng.forward(b, training=False))
        for k in ("mel", "stop", "alignment1"):
            e1 = rel_err(outt[k].detach().cpu().numpy(), reft[k].numpy())
            e2 = rel_err(outt[k].detach().cpu().numpy(), fw[k].detach().float().cpu().numpy())
            print("teacher-fed", k, e1, e2)
            assert e1 < 5e-4 and e2 < 2e-5, (k, e1, e2)
        # (iii) the masks are really applied
        cfg0, _ = make_params(cfg_kw, seed=2)
        eng0, _ = make_engine(cfg0, P)
        for name, (m, v) in eng.bn.items():
            eng0.bn[name][0].copy_(m); eng0.bn[name][1].copy_(v)
        plain = infer(eng0, bt["source"], bt["source_length"], max_steps=steps, min_steps=10 ** 6, **espk)
        assert rel_err(plain["mel"].detach().cpu().numpy(), out["mel"].detach().cpu().numpy()) > 1e-3
        # (iv) without a pinned seed every synthesis call draws fresh masks (the reference's stateful RNG gives output
        # variation from run to run - the point of the flag); a pinned seed reproduces
        r1 = infer(eng, bt["source"], bt["source_length"], max_steps=steps, min_steps=10 ** 6, **espk)["mel"].detach().cpu().numpy()
        r2 = infer(eng, bt["source"], bt["source_length"], max_steps=steps, min_steps=10 ** 6, **espk)["mel"].detach().cpu().numpy()
        r3 = infer(eng, bt["source"], bt["source_length"], max_steps=steps, min_steps=10 ** 6, dropout_seed=7, **espk)["mel"]
        assert rel_err(r1, r2) > 1e-3
        assert np.array_equal(r3.detach().cpu().numpy(), out["mel"].detach().cpu().numpy())
    finally:
        DecodeSession.FUSE = old


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("cfg_kw,hops,B,Ti,steps", [(MEDIUM, (2, 3), 4, 33, 14), (SMALL, (3, 2), 3, 9, 12)])
def test_multi_hop_decode(cfg_kw, hops, B, Ti, steps, fuse):
    """stacked SelfAttentionTransformer blocks while synthesising (TransformerWrapper re-runs every block over the history,
    modules/rnn_wrappers.py:87-124): one K|V|Q cache per hop; free run against the oracle, teacher-fed pass against the
    batched evaluation forward"""
    from oracle import torch_ref
    from satt_amd.inference import DecodeSession, infer
    kw = dict(cfg_kw, sa_num_hop=hops[0], dec_sa_num_hop=hops[1])
    cfg, P = make_params(kw, seed=2)
    batch = small_batch(cfg, B, Ti, steps * cfg.r, seed=5)
    eng, mv = make_engine(cfg, P)
    ocfg = torch_ref.Cfg(**kw)
    Pt = torch_ref.to_torch(P)
    bt = torch_ref.batch_to_torch(batch)
    old = DecodeSession.FUSE
    DecodeSession.FUSE = fuse
    try:
        ref = torch_ref.infer(Pt, bt["source"], bt["source_length"], ocfg, steps, mv, min_steps=10 ** 6)
        out = infer(eng, bt["source"], bt["source_length"], max_steps=steps, min_steps=10 ** 6)
        for k in ("mel", "stop", "alignment1", "alignment2"):
            e = rel_err(out[k].detach().cpu().numpy(), ref[k].numpy())
            print(k, e)
            assert e < 5e-4, (k, e)
        outt = infer(eng, bt["source"], bt["source_length"], teacher=bt["mel"].float())
        fw = eng.outputs(eng.forward(eng.to_device_batch(batch), training=False))
        for k in ("mel", "stop", "alignment1"):
            e = rel_err(outt[k].detach().cpu().numpy(), fw[k].detach().float().cpu().numpy())
            assert e < 2e-5, (k, e)
    finally:
        DecodeSession.FUSE = old


def test_free_running_stop_rule():
    """the stop rule fires (all samples, t > min_steps) and the returned length reflects it"""
    from satt_amd.inference import infer
    cfg, P = make_params(SMALL, seed=2)
    P = dict(P)
    b = np.array(P["dec.out.b"], dtype=np.float32).copy(); b[-1] = 50.0          # stop logit always large
    P["dec.out.b"] = b
    batch = small_batch(cfg, 3, 9, 12, seed=5)
    eng, _ = make_engine(cfg, P)
    out = infer(eng, batch["source"], batch["source_length"], max_steps=30, min_steps=4)
    assert out["steps"] == 6          # first step with t > 4 is t = 5 -> 6 steps run
    assert out["mel"].shape == (3, 6 * cfg.r, cfg.num_mels)


@pytest.mark.parametrize("cfg_kw,B,Ti,steps", [(MEDIUM, 4, 33, 14), (SMALL, 3, 9, 12), (dict(), 2, 21, 8)])
def test_forced_alignment_mode(cfg_kw, B, Ti, steps):
    """use_forced_alignment_mode (modules/teacher_forcing_attention.py:13-78, models/models.py:411-428):
    (1) against the float64 oracle with arbitrary (row-normalised, length-masked) teacher alignments;
    (2) property: feeding a free run's own alignments back reproduces that run (the mechanisms are bypassed, every
        other part of the step is unchanged)."""
    from oracle import torch_ref
    from satt_amd.inference import infer
    cfg, P = make_params(cfg_kw, seed=2)
    batch = small_batch(cfg, B, Ti, 2 * cfg.r, seed=5)
    eng, mv = make_engine(cfg, P)
    ocfg = torch_ref.Cfg(**cfg_kw)
    Pt = torch_ref.to_torch(P)
    src, sl = torch.as_tensor(batch["source"]), torch.as_tensor(batch["source_length"])
    g = torch.Generator().manual_seed(3)
    mask = (torch.arange(Ti)[None, None, :] < sl[:, None, None]).double()
    ta = []
    for _ in range(2):
        a = torch.rand(B, steps, Ti, generator=g, dtype=torch.float64) * mask
        ta.append(a / a.sum(-1, keepdim=True))
    ref = torch_ref.infer(Pt, src, sl, ocfg, steps, mv, min_steps=10 ** 6, teacher_alignments=ta)
    out = infer(eng, src, sl, max_steps=steps, min_steps=10 ** 6, teacher_alignments=ta)
    for k in ("mel", "stop", "alignment1", "alignment2"):
        e = rel_err(out[k].detach().cpu().numpy(), ref[k].numpy())
        print(k, e)
        assert e < 5e-4, (k, e)
    free = infer(eng, src, sl, max_steps=steps, min_steps=10 ** 6)
    again = infer(eng, src, sl, max_steps=steps, min_steps=10 ** 6,
                  teacher_alignments=(free["alignment1"], free["alignment2"]))
    for k in ("mel", "stop"):
        e = rel_err(again[k].detach().cpu().numpy(), free[k].detach().cpu().numpy())
        print("self", k, e)
        assert e < 2e-5, (k, e)


@pytest.mark.parametrize("cfg_kw,B,Ti,Tm", [(MEDIUM, 4, 33, 28), (SMALL, 3, 9, 12)])
def test_eval_double_pass_matches_oracle(cfg_kw, B, Ti, Tm):
    """EVAL mode of the reference's model_fn (models/models.py:517-564): free run over exactly Td steps + teacher-fed
    validation pass, both scored with the training losses"""
    from oracle import torch_ref
    from satt_amd.inference import evaluate
    cfg, P = make_params(cfg_kw, seed=2)
    batch = small_batch(cfg, B, Ti, Tm, seed=5)
    eng, mv = make_engine(cfg, P)
    ocfg = torch_ref.Cfg(**cfg_kw)
    Pt = torch_ref.to_torch(P)
    bt = torch_ref.batch_to_torch(batch)
    Td = Tm // cfg.r
    free = torch_ref.infer(Pt, bt["source"], bt["source_length"], ocfg, Td, mv, min_steps=10 ** 9)
    tf = torch_ref.infer(Pt, bt["source"], bt["source_length"], ocfg, None, mv, teacher=bt["mel"])
    ev = evaluate(eng, batch)
    for name, o in (("", free), ("_with_teacher", tf)):
        ml, dl = torch_ref.losses(o["mel"], o["stop"], bt)
        for k, v in (("mel_loss", ml), ("done_loss", dl)):
            got, want = ev[k + name], float(v)
            print(k + name, got, want)
            assert abs(got - want) < 5e-4 * max(1.0, abs(want)), (k + name, got, want)
    assert abs(ev["loss"] - (ev["mel_loss"] + ev["done_loss"])) < 1e-5


@pytest.mark.parametrize("cfg_kw,B,Ti,steps", [(MEDIUM, 4, 33, 14), (SMALL, 3, 9, 12)])
def test_baseline_tacotron_decode(cfg_kw, B, Ti, steps):
    """ExtendedTacotronV1Model (one attention source, no decoder self-attention: reference modules/module.py:530-623):
    free-running decode against the float64 oracle, and the teacher-fed step-by-step pass against the batched forward"""
    from oracle import torch_ref
    from satt_amd.inference import infer
    kw = dict(cfg_kw, sa_units=0, att2_units=0, dec_sa_units=0, att1_units=cfg_kw["att_rnn_units"])
    cfg, P = make_params(kw, seed=2)
    batch = small_batch(cfg, B, Ti, steps * cfg.r, seed=5)
    eng, mv = make_engine(cfg, P)
    Pt = torch_ref.to_torch(P)
    src, sl = torch.as_tensor(batch["source"]), torch.as_tensor(batch["source_length"])
    ref = torch_ref.infer(Pt, src, sl, torch_ref.Cfg(**kw), steps, mv, min_steps=10 ** 6)
    out = infer(eng, src, sl, max_steps=steps, min_steps=10 ** 6)
    assert out["steps"] == ref["steps"] == steps and out["sa_out"] is None
    for k in ("mel", "stop", "alignment1"):
        e = rel_err(out[k].detach().cpu().numpy(), ref[k].numpy())
        print(k, e)
        assert e < 5e-4, (k, e)
    b = eng.to_device_batch(batch)
    fwd = {k: v.detach().float().cpu().numpy() for k, v in eng.outputs(eng.forward(b, training=False)).items()}
    val = infer(eng, b["source"], b["source_length"], teacher=b["mel"])
    for k in ("mel", "stop", "alignment1"):
        assert rel_err(val[k].detach().cpu().numpy(), fwd[k]) < 2e-5, k
    # forced alignments of the single mechanism (teacher_forcing_forward, models/models.py:62-77)
    forced = infer(eng, src, sl, max_steps=steps, min_steps=10 ** 6, teacher_alignments=(out["alignment1"], None))
    assert rel_err(forced["mel"].cpu().numpy(), out["mel"].cpu().numpy()) < 1e-4


def test_captured_graph_equals_step_by_step_launches():
    """the hipGraph replay (several steps per graph, device-side step counter and stop flag) and the same kernels issued
    one by one produce bit-identical outputs, and the stop rule ends both at the same step"""
    from satt_amd.inference import infer
    cfg, P = make_params(MEDIUM, seed=4)
    P = dict(P)
    b = np.array(P["dec.out.b"], dtype=np.float32).copy(); b[-1] = 50.0          # stop logit always large
    batch = small_batch(cfg, 4, 33, 12, seed=6)
    eng, _ = make_engine(cfg, P)
    kw = dict(max_steps=26, min_steps=10 ** 6, check_every=4)
    g = infer(eng, batch["source"], batch["source_length"], use_graph=True, **kw)
    e = infer(eng, batch["source"], batch["source_length"], use_graph=False, **kw)
    assert g["steps"] == e["steps"] == 26
    for k in ("mel", "stop", "alignment1", "alignment2"):
        assert torch.equal(g[k], e[k]), k
    again = infer(eng, batch["source"], batch["source_length"], use_graph=True, **kw)      # the cached session, reset
    assert torch.equal(again["mel"], g["mel"])
    P["dec.out.b"] = b
    eng2, _ = make_engine(cfg, P)
    for ug in (True, False):
        out = infer(eng2, batch["source"], batch["source_length"], max_steps=40, min_steps=9, check_every=4, use_graph=ug)
        assert out["steps"] == 11 and out["mel"].shape[1] == 11 * cfg.r, (ug, out["steps"])


@pytest.mark.parametrize("cfg_kw,B,Ti,steps", [(MEDIUM, 4, 33, 14), (dict(), 2, 21, 10)])
def test_transition_agent_decode(cfg_kw, B, Ti, steps):
    """use_forward_attention_transition_agent at inference: free run against the float64 oracle, teacher-fed pass against the
    batched forward of the training kernels (two independent implementations of the per-step prediction of u)"""
    from oracle import torch_ref
    from satt_amd.inference import infer
    kw = dict(cfg_kw, transition_agent=True)
    cfg, P = make_params(kw, seed=2)
    P["dec.att1.Wa"] = (3.0 * P["dec.att1.Wa"]).astype(np.float32)
    batch = small_batch(cfg, B, Ti, steps * cfg.r, seed=5)
    eng, mv = make_engine(cfg, P)
    Pt = torch_ref.to_torch(P)
    src, sl = torch.as_tensor(batch["source"]), torch.as_tensor(batch["source_length"])
    ref = torch_ref.infer(Pt, src, sl, torch_ref.Cfg(**kw), steps, mv, min_steps=10 ** 6)
    out = infer(eng, src, sl, max_steps=steps, min_steps=10 ** 6)
    plain = torch_ref.infer({k: v for k, v in Pt.items() if k not in ("dec.att1.Wa", "dec.att1.ba")}, src, sl,
                            torch_ref.Cfg(**cfg_kw), steps, mv, min_steps=10 ** 6)
    assert float((ref["alignment1"] - plain["alignment1"]).abs().max()) > 1e-3       # the agent matters on this input
    for k in ("mel", "stop", "alignment1", "alignment2"):
        e = rel_err(out[k].detach().cpu().numpy(), ref[k].numpy())
        print(k, e)
        assert e < 5e-4, (k, e)
    b = eng.to_device_batch(batch)
    fwd = {k: v.detach().float().cpu().numpy() for k, v in eng.outputs(eng.forward(b, training=False)).items()}
    val = infer(eng, b["source"], b["source_length"], teacher=b["mel"])
    for k in ("mel", "stop", "alignment1", "alignment2"):
        assert rel_err(val[k].detach().cpu().numpy(), fwd[k]) < 2e-5, k


@pytest.mark.parametrize("cfg_kw,B,Ti,steps", [(dict(), 1, 100, 24), (dict(), 8, 57, 16), (MEDIUM, 5, 33, 14),
                                               (dict(sa_units=0, att2_units=0, dec_sa_units=0, att1_units=256), 2, 40, 12)])
def test_bf16_decode_chained_launches(cfg_kw, B, Ti, steps):
    """The benchmark precision of config 5 (bf16 weight shadows): (1) the chained launches (pre-net 0 -> pre-net 1 ->
    attention LSTM; output transform -> mel | stop: csrc/decode.hip dec_chain_k / dec_linear2_k) against the same layers
    launched one by one - same bf16 weights, fp32 accumulation in a different order; (2) the bf16 run against the fp32 run
    of the same engine - the rounding of the weights only."""
    from satt_amd import ops
    from satt_amd.inference import infer, DecodeSession
    cfg, P = make_params(cfg_kw, seed=4)
    batch = small_batch(cfg, B, Ti, 12, seed=6)
    eng, _ = make_engine(cfg, P, "bf16")
    kw = dict(max_steps=steps, min_steps=10 ** 6)
    DecodeSession.MEGA = False           # (this test is about the launch-per-layer path)
    try:
        fused = infer(eng, batch["source"], batch["source_length"], **kw)
        ses = eng._decode_sessions[next(reversed(eng._decode_sessions))]
        n_fused = ses.kernel_launches
        DecodeSession.MAX_CHAIN = 2          # the two-layer chain (pre-net 0 -> pre-net 1 -> attention LSTM) as well
        deep = infer(eng, batch["source"], batch["source_length"], **kw)
        assert eng._decode_sessions[next(reversed(eng._decode_sessions))].kernel_launches < n_fused
        assert rel_err(deep["mel"].cpu().numpy(), fused["mel"].cpu().numpy()) < 1e-5
        DecodeSession.FUSE = False
        plain = infer(eng, batch["source"], batch["source_length"], **kw)
        n_plain = eng._decode_sessions[next(reversed(eng._decode_sessions))].kernel_launches
    finally:
        DecodeSession.FUSE = True
        DecodeSession.MAX_CHAIN = 1
        DecodeSession.MEGA = True
    assert n_fused < n_plain, (n_fused, n_plain)          # the chains were actually taken
    keys = ["mel", "stop", "alignment1"] + (["alignment2"] if cfg.dual else [])
    for k in keys:
        e = rel_err(fused[k].cpu().numpy(), plain[k].cpu().numpy())
        print("fused vs plain", k, e)
        assert e < 1e-5, (k, e)          # fp32 sums in a different order, fed back through `steps` recurrent steps
    ops.set_precision("f32")
    eng.refresh_shadows()
    ref = infer(eng, batch["source"], batch["source_length"], **kw)
    ops.set_precision("bf16")
    for k in keys:
        e = rel_err(fused[k].cpu().numpy(), ref[k].cpu().numpy())
        print("bf16 vs f32", k, e)
        assert e < 2e-2, (k, e)
    assert torch.isfinite(fused["mel"]).all()


@pytest.mark.parametrize("form,mode,B,Ti,steps",
                         [("tables", m, *c) for c in [(1, 100, 24), (2, 57, 19), (2, 160, 16), (1, 33, 9), (1, 140, 10)] for m in ("free", "teacher", "stop")] +
                         [("nofold", "free", 1, 100, 40), ("nofold", "stop", 2, 57, 19),      # (r6: without the folded feedback - the r5 step)
                          ("tables32", "free", 1, 100, 40), ("tables32", "stop", 2, 57, 40), ("tables32", "teacher", 1, 140, 33),
                          # the shapes' edges: the longest memory, a memory shorter than the workgroup count, and more than 512 steps
                          # (key chunks grow beyond the 32 LDS-resident rows: the cached self-attention reads the cache again)
                          ("tables32", "free", 1, 256, 12), ("tables32", "free", 2, 7, 12), ("tables32", "free", 1, 60, 530)])
def test_persistent_decode_kernel_equals_the_launch_per_layer_path(form, mode, B, Ti, steps):
    """The persistent step kernel (one launch per 8 decoder steps: csrc/decode_mega2.hip - register-resident weights, granule
    exchanges, contexts folded into per-utterance tables; LDS-resident tables at Ti <= 112, global ones above, B = 1 and 2; r6: all
    workgroups on one XCD, the mel projection folded into the next step's first pre-net layer - with MEGA_STEPS = 8 every eighth
    step takes the unfolded form) against the hipGraph of launch-per-layer
    steps it replaces: same bf16 weight shadows, same buffers, fp32 sums in a different order - through `steps` recurrent steps
    (several launches, a ragged last one), free-running, teacher-fed and with the stop rule firing."""
    from satt_amd.inference import infer, DecodeSession
    cfg, P = make_params(dict(), seed=4)
    P = dict(P)
    if mode == "stop":
        b = np.array(P["dec.out.b"], dtype=np.float32).copy(); b[-1] = 50.0          # stop logit always large
        P["dec.out.b"] = b
    batch = small_batch(cfg, B, Ti, steps * cfg.r, seed=6)
    eng, _ = make_engine(cfg, P, "bf16")
    kw = dict(teacher=torch.as_tensor(batch["mel"])) if mode == "teacher" else \
        dict(max_steps=steps, min_steps=(5 if mode == "stop" else 10 ** 6))
    try:
        DecodeSession.MEGA = True
        DecodeSession.MEGA_STEPS = 32 if form == "tables32" else 8      # (8: several launches and a ragged last one within few steps)
        DecodeSession.MEGA_FOLD_FEEDBACK = form != "nofold"
        new = infer(eng, batch["source"], batch["source_length"], **kw)
        ses = eng._decode_sessions[next(reversed(eng._decode_sessions))]
        assert ses.mega is not None and ses.kernel_launches == 1          # the persistent kernel was actually taken
        again = infer(eng, batch["source"], batch["source_length"], **kw)  # the cached session, reset
        DecodeSession.MEGA = False
        old = infer(eng, batch["source"], batch["source_length"], **kw)
        assert eng._decode_sessions[next(reversed(eng._decode_sessions))].mega is None
    finally:
        DecodeSession.MEGA = True
        DecodeSession.MEGA_STEPS = 128
        DecodeSession.MEGA_FOLD_FEEDBACK = True
    assert new["steps"] == old["steps"] == (7 if mode == "stop" else steps)
    for k in ("mel", "stop", "alignment1", "alignment2"):
        e = rel_err(new[k].cpu().numpy(), old[k].cpu().numpy())
        print(mode, k, e)
        assert e < 2e-5, (k, e)
        assert torch.equal(new[k], again[k]), k
    assert torch.isfinite(new["mel"]).all()


@pytest.mark.parametrize("B,Ti", [(1, 100), (2, 57)])
def test_persistent_decode_kernel_hands_over_to_the_launch_per_layer_path_and_back(B, Ti):
    """include/satt_hip.h: the caller may switch between satt_dec_mega and the launch-per-layer entry points from one LAUNCH to the
    next - same buffers.  8 steps persistent, 8 steps launch per layer, 8 steps persistent == 24 steps persistent: recurrent state,
    contexts (written by the persistent kernel at the last step of a launch only), location-conv input, forward variable, step
    counters, K|V|Q cache rows and histories all cross the boundary in both directions."""
    from satt_amd.inference import infer, DecodeSession
    cfg, P = make_params(dict(), seed=4)
    K, steps = 8, 24
    batch = small_batch(cfg, B, Ti, steps * cfg.r, seed=6)
    eng, _ = make_engine(cfg, dict(P), "bf16")
    try:
        DecodeSession.MEGA_STEPS = K
        ref = infer(eng, batch["source"], batch["source_length"], max_steps=steps, min_steps=10 ** 6)
    finally:
        DecodeSession.MEGA_STEPS = 128
    ses = eng._decode_sessions[next(reversed(eng._decode_sessions))]
    assert ses.mega is not None and ses.K == K
    ses.reset()                      # (memories, context tables and folded weights of the utterance stay in place)
    ses.replay()
    for _ in range(K):
        ses.run_step()
    ses.replay()
    torch.cuda.synchronize()
    ses.check()
    NO = ses.yout.shape[-1]
    for name, a, b in (("frames", ses.yout[:, 1:steps + 1].reshape(B * steps, NO), ref["yout"]), ("alignment1", ses.al1[:, :steps], ref["alignment1"]),
                       ("alignment2", ses.al2[:, :steps], ref["alignment2"])):
        e = rel_err(a.cpu().numpy(), b.cpu().numpy())
        print(name, e)
        assert e < 2e-5, (name, e)

