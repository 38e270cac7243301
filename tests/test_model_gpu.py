"""GPU parity of the full teacher-forced training path against the float64 oracle (dropout / zoneout ON with
identical counter-based masks).  precision='f32' (exact fp32 MFMA) is held to fp32 tolerance; precision='bf16'
(the benchmark dtype) to the mel-L1 1e-3 bar of BASELINE.json plus loose element-wise bounds."""
import numpy as np
import pytest
import torch

from common import MEDIUM, SMALL, make_params, oracle_run, rel_err, small_batch

pytestmark = pytest.mark.gpu


def run_engine(cfg, P, batch, seed, prec, dalign=None, clusters=True):
    from satt_amd import ops
    from satt_amd.engine import Engine
    ops.set_precision(prec)
    eng = Engine(cfg, "cuda", params=P, rng_seed=seed)
    eng.use_clusters = clusters
    b = eng.to_device_batch(batch)
    eng.zero_grad()
    ctx = eng.forward(b, training=True)
    if dalign is not None:
        ctx["dalign1"] = torch.as_tensor(dalign[0], dtype=torch.float32, device="cuda").contiguous()
        ctx["dalign2"] = torch.as_tensor(dalign[1], dtype=torch.float32, device="cuda").contiguous()
    eng.backward(ctx)
    torch.cuda.synchronize()
    eng.check_clusters(ctx)
    out = {k: v.detach().float().cpu().numpy() for k, v in eng.outputs(ctx).items()}
    grads = {k: v.detach().cpu().numpy() for k, v in eng.G.items()}
    eng.last_ctx = ctx
    return eng, out, grads


def assert_same_xcd_fast_path(eng, B):
    """every cluster workgroup of the step's LAST attention launch (the backward kernel) and of the LSTM backward pass
    exchanged through same-XCD plain stores (csrc/cluster_xchg.h) - read back from the kernels' own handshake result"""
    from satt_amd import ops
    ctx = eng.last_ctx
    Ca, aws = ctx["att_cluster"]
    assert Ca > 0, "attention cluster kernels were not selected"
    # the counters are sticky (launches never clear the workspace tail): every workgroup-launch since the engine allocated
    # the workspace is counted, so "all on the fast path" = none on the slow one
    n, slow = ops.attn_cluster_fastpath(ctx["att_params"], Ca, aws)
    assert slow == 0 and n > 0 and n % (B * Ca) == 0, ("attention cluster: %d fast, %d slow workgroup-launches" % (n, slow))
    Cn, cws1, cws2 = ctx["cluster"]
    assert Cn > 0
    for ws in (cws1, cws2):
        m, slow = ops.lstm_cluster_fastpath(ws, B, eng.cfg.dec_units, Cn)
        assert slow == 0 and m > 0 and m % (B * Cn) == 0, ("LSTM cluster: %d fast, %d slow workgroup-launches" % (m, slow))


def report(out, ref, grads, gref, keys):
    rows = []
    for k in keys:
        rows.append((k, rel_err(out[k], ref[k].detach().numpy() if hasattr(ref[k], "detach") else ref[k])))
    for k in grads:
        rows.append(("grad:" + k, rel_err(grads[k], gref[k])))
    for k, e in rows:
        print("%-28s rel_err=%.3e" % (k, e))
    return dict(rows)


@pytest.mark.parametrize("clusters", [True, False])
@pytest.mark.parametrize("cfg_kw,B,Ti,Tm", [(SMALL, 3, 9, 12), (MEDIUM, 5, 37, 46)])
def test_f32_parity_forward_backward(cfg_kw, B, Ti, Tm, clusters):
    cfg, P = make_params(cfg_kw, seed=1)
    batch = small_batch(cfg, B, Ti, Tm, seed=3)
    g = np.random.default_rng(0)
    Td = Tm // cfg.r
    dal = (g.normal(0, 0.05, (B, Td, Ti)), g.normal(0, 0.05, (B, Td, Ti)))
    ref, col, gref = oracle_run(cfg_kw, P, batch, True, seed=7, dalign=dal)
    eng, out, grads = run_engine(cfg, P, batch, 7, "f32", dalign=dal, clusters=clusters)
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref,
                  ["lstm_out", "sa_out", "alignment1", "alignment2", "dec_out", "mel", "stop", "loss", "mel_loss",
                   "done_loss"])
    bad = {k: e for k, e in errs.items() if not (e < 2e-4)}
    assert not bad, bad


@pytest.mark.parametrize("cfg_kw,Ti", [(dict(), 21), (dict(att_kernel=6), 21), (dict(), 300)])
def test_f32_parity_production_dims(cfg_kw, Ti):
    """The LJSpeech configuration itself (BASELINE configs[1] dimensions, short sequences): this is the shape the
    cluster kernels are specialised for at compile time (and, with a different filter width, the generic build of the
    same register layout), so the specialised code paths are held to the same fp32 bar as the small configurations."""
    B, Tm = 2, 24          # Ti = 300: 75 memory rows per cluster member (several row passes, 3 value K tiles)
    cfg, P = make_params(cfg_kw, seed=5)
    batch = small_batch(cfg, B, Ti, Tm, seed=6)
    g = np.random.default_rng(1)
    Td = Tm // cfg.r
    dal = (g.normal(0, 0.05, (B, Td, Ti)), g.normal(0, 0.05, (B, Td, Ti)))
    ref, col, gref = oracle_run(cfg_kw, P, batch, True, seed=9, dalign=dal)
    eng, out, grads = run_engine(cfg, P, batch, 9, "f32", dalign=dal, clusters=True)
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref,
                  ["lstm_out", "sa_out", "alignment1", "alignment2", "dec_out", "mel", "stop", "loss", "mel_loss",
                   "done_loss"])
    # with 600 encoder positions x 2048 bank channels a handful of ReLU / max-pool decisions sit within fp32 rounding
    # of a tie (float64 oracle vs fp32 summation order; the split-K forward convolutions accumulate with atomics, so
    # which ones flip varies from run to run): those flip single gradient entries of the encoder front end / CBHG,
    # so at Ti = 300 the encoder-side gradients are judged by relative L2 error and everything else stays max-norm
    front = ("grad:embedding", "grad:enc.prenet", "grad:enc.bank", "grad:enc.proj", "grad:enc.highway")
    bad = {k: e for k, e in errs.items() if not (e < 2e-4) and not (Ti > 100 and k.startswith(front))}
    if Ti > 100:
        for k in grads:
            if ("grad:" + k).startswith(front):
                a, b = grads[k].astype(np.float64).ravel(), gref[k].astype(np.float64).ravel()
                l2 = float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
                if not l2 < 5e-3:
                    bad["l2:" + k] = l2
    assert not bad, bad


@pytest.mark.parametrize("cfg_kw,B,Ti,Tm", [(MEDIUM, 8, 37, 46), (MEDIUM, 16, 29, 34), (dict(), 8, 21, 24),
                                             (dict(), 16, 21, 24), (dict(), 32, 21, 16)])
def test_f32_parity_same_xcd_exchange(cfg_kw, B, Ti, Tm):
    """B % 8 == 0 puts the C members of every sample's cluster on one XCD, and the kernels then publish their granules
    with PLAIN stores (cluster_xchg.h:17-21) - the exchange path of the benchmark (B = 32).  Same oracle comparison as
    above (every forward tensor, all parameter gradients, dropout / zoneout on), at MEDIUM dims and at the LJSpeech
    dims, plus the proof that the fast path produced these numbers."""
    cfg, P = make_params(cfg_kw, seed=1)
    batch = small_batch(cfg, B, Ti, Tm, seed=3)
    g = np.random.default_rng(0)
    Td = Tm // cfg.r
    dal = (g.normal(0, 0.05, (B, Td, Ti)), g.normal(0, 0.05, (B, Td, Ti)))
    ref, col, gref = oracle_run(cfg_kw, P, batch, True, seed=7, dalign=dal)
    eng, out, grads = run_engine(cfg, P, batch, 7, "f32", dalign=dal, clusters=True)
    assert_same_xcd_fast_path(eng, B)
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref,
                  ["lstm_out", "sa_out", "alignment1", "alignment2", "dec_out", "mel", "stop", "loss", "mel_loss",
                   "done_loss"])
    bad = {k: e for k, e in errs.items() if not (e < 2e-4)}
    assert not bad, bad


def test_bf16_same_xcd_exchange_production_dims():
    """benchmark precision through the plain-store exchange at the LJSpeech dims, B = 16, against the oracle"""
    cfg_kw, B, Ti, Tm = dict(), 16, 21, 24
    cfg, P = make_params(cfg_kw, seed=2)
    batch = small_batch(cfg, B, Ti, Tm, seed=4)
    ref, col, gref = oracle_run(cfg_kw, P, batch, True, seed=11)
    eng, out, grads = run_engine(cfg, P, batch, 11, "bf16")
    assert_same_xcd_fast_path(eng, B)
    assert abs(float(out["mel_loss"]) - float(ref["mel_loss"].detach())) < 1e-3
    assert rel_err(out["mel"], ref["mel"].detach().numpy()) < 5e-2
    assert rel_err(out["alignment1"], ref["alignment1"].detach().numpy()) < 5e-2
    for k in grads:
        a, b = grads[k].astype(np.float64).ravel(), gref[k].astype(np.float64).ravel()
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        assert cos > 0.98, (k, cos)


@pytest.mark.parametrize("attention,cumulative,vsum", [("forward", False, None), ("location_sensitive", False, None), ("forward", True, None),
                                                       ("location_sensitive", True, None), ("forward", False, 35.0), ("forward", False, 45.0)])
def test_bf16_folded_forward_kernel_branches(attention, cumulative, vsum):
    """The folded bf16 forward kernel (csrc/attn_cluster.hip, FOLD) has three forms of its normalisation: LAZY (r5: the forward
    variable carried un-normalised, nothing but the reciprocals of the sums on the chain; needs sum|v| <= 30 and a non-cumulative
    location input), the in-chain form with the constant softmax shift (sum|v| <= 40, or cumulative weights), and the in-chain form
    with member-local maxima (sum|v| > 40).  `attention=location_sensitive` runs any of them with the unit forward weight.  Each
    against the float64 oracle at the LJSpeech dims, benchmark precision, B = 16 (same-XCD exchange), dropout / zoneout on."""
    cfg_kw, B, Ti, Tm = dict(attention=attention, cumulative_weights=cumulative), 16, 21, 24
    cfg, P = make_params(cfg_kw, seed=2)
    if vsum is not None:
        P = dict(P)
        for k in ("dec.att1.v", "dec.att2.v"):
            v = np.array(P[k], dtype=np.float64)
            P[k] = (v * (vsum / np.abs(v).sum())).astype(np.float32)
    batch = small_batch(cfg, B, Ti, Tm, seed=4)
    ref, col, gref = oracle_run(cfg_kw, P, batch, True, seed=11)
    eng, out, grads = run_engine(cfg, P, batch, 11, "bf16")
    assert "saf" in eng.last_ctx, "the folded forward kernel was not selected"
    assert_same_xcd_fast_path(eng, B)
    assert abs(float(out["mel_loss"]) - float(ref["mel_loss"].detach())) < 1e-3
    assert rel_err(out["mel"], ref["mel"].detach().numpy()) < 5e-2
    assert rel_err(out["alignment1"], ref["alignment1"].detach().numpy()) < 5e-2
    assert rel_err(out["alignment2"], ref["alignment2"].detach().numpy()) < 5e-2
    for k in grads:
        a, b = grads[k].astype(np.float64).ravel(), gref[k].astype(np.float64).ravel()
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        assert cos > 0.98, (k, cos)


def test_precision_switch_on_a_live_engine_keeps_its_cluster_workspace():
    """The exchange workspace of the attention clusters is cached per shape, not per precision: the bf16 kernels' fixed granule
    layout (Ti = 160 whatever the launch's Ti) must fit the workspace an engine allocated in f32 mode - satt_attn_cluster_ws_bytes
    is a function of the specialisation alone.  One engine, f32 step first, then bf16: the bf16 step equals a fresh bf16 engine's."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    cfg, P = make_params(dict(), seed=2)
    batch = small_batch(cfg, 8, 21, 24, seed=4)
    losses = {}
    try:
        ops.set_precision("f32")
        eng = Engine(cfg, "cuda", params=P, rng_seed=11)
        b = eng.to_device_batch(batch)
        for prec in ("f32", "bf16"):
            ops.set_precision(prec)
            eng.zero_grad()
            ctx = eng.forward(b, training=True)
            eng.backward(ctx)
            torch.cuda.synchronize()
            eng.check_clusters(ctx)
            losses[prec] = (float(eng.losses[2]), eng.grad.clone())
        fresh = Engine(cfg, "cuda", params=P, rng_seed=11)
        fb = fresh.to_device_batch(batch)
        fresh.zero_grad()
        ctx = fresh.forward(fb, training=True)
        fresh.backward(ctx)
        torch.cuda.synchronize()
        fresh.check_clusters(ctx)
    finally:
        ops.set_precision("bf16")
    assert abs(losses["bf16"][0] - float(fresh.losses[2])) < 1e-6
    assert float((losses["bf16"][1] - fresh.grad).abs().max()) <= 1e-6 * float(fresh.grad.abs().max())
    assert abs(losses["bf16"][0] - losses["f32"][0]) < 5e-3


def test_f32_parity_large_energy_bound():
    """sum|v| > 40 switches the cluster forward kernel from the constant-shift softmax numerators to the
    member-local-max path (attn_cluster.hip, phase 6): both must match the oracle."""
    cfg_kw, B, Ti, Tm = MEDIUM, 5, 37, 46
    cfg, P = make_params(cfg_kw, seed=1)
    P = dict(P)
    for k in ("dec.att1.v", "dec.att2.v"):
        v = np.array(P[k], dtype=np.float64)
        P[k] = (v * (45.0 / np.abs(v).sum())).astype(np.float32)
    batch = small_batch(cfg, B, Ti, Tm, seed=3)
    ref, col, gref = oracle_run(cfg_kw, P, batch, True, seed=7)
    eng, out, grads = run_engine(cfg, P, batch, 7, "f32", clusters=True)
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref,
                  ["alignment1", "alignment2", "dec_out", "mel", "loss"])
    bad = {k: e for k, e in errs.items() if not (e < 5e-4)}
    assert not bad, bad


def test_f32_parity_postnet_v2():
    """optional PostNetV2 conv stack + its extra spec_loss term (reference models/models.py:92-100,116-118)"""
    cfg_kw = dict(SMALL, use_postnet_v2=True, num_postnet_v2_layers=3, postnet_v2_kernel_size=5,
                  postnet_v2_out_channels=16, postnet_v2_drop_rate=0.5)
    B, Ti, Tm = 3, 9, 12
    cfg, P = make_params(cfg_kw, seed=1)
    batch = small_batch(cfg, B, Ti, Tm, seed=3)
    ref, col, gref = oracle_run(cfg_kw, P, batch, True, seed=7)
    eng, out, grads = run_engine(cfg, P, batch, 7, "f32", clusters=True)
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref,
                  ["mel", "mel_postnet", "postnet_mel_loss", "loss", "stop"])
    bad = {k: e for k, e in errs.items() if not (e < 2e-4)}
    assert not bad, bad


@pytest.mark.parametrize("cfg_kw,ns,B", [(MEDIUM, 7, 4), (dict(), 152, 8)])
def test_f32_parity_multi_speaker_vctk(cfg_kw, ns, B):
    """BASELINE configs[3]: speaker embedding -> MultiSpeakerPreNet (reference modules/multi_speaker_modules.py); the second
    case is examples/vctk/self-attention-tacotron.json itself (152 speakers from id 225, LJSpeech layer sizes)."""
    cfg_kw = dict(cfg_kw, num_speakers=ns, speaker_dim=16, speaker_offset=225)
    cfg, P = make_params(cfg_kw, seed=4)
    batch = small_batch(cfg, B, 21, 26, seed=8)
    batch["speaker_id"] = (np.random.default_rng(1).integers(0, ns, B) + 225).astype(np.int64)
    ref, col, gref = oracle_run(cfg_kw, P, batch, True, seed=13)
    eng, out, grads = run_engine(cfg, P, batch, 13, "f32")
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref, ["mel", "stop", "alignment1", "loss"])
    bad = {k: e for k, e in errs.items() if not (e < 2e-4)}
    assert not bad, bad
    assert float(np.abs(grads["speaker_embedding"]).max()) > 0


@pytest.mark.parametrize("cfg_kw,B,Ti,Tm", [(MEDIUM, 5, 37, 46), (dict(), 2, 21, 24)])
def test_bf16_parity(cfg_kw, B, Ti, Tm):
    cfg, P = make_params(cfg_kw, seed=2)
    batch = small_batch(cfg, B, Ti, Tm, seed=4)
    ref, col, gref = oracle_run(cfg_kw, P, batch, True, seed=11)
    eng, out, grads = run_engine(cfg, P, batch, 11, "bf16")
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref,
                  ["lstm_out", "sa_out", "alignment1", "alignment2", "dec_out", "mel", "stop", "loss", "mel_loss",
                   "done_loss"])
    # BASELINE.json: mel L1 within 1e-3 of the reference semantics
    assert abs(float(out["mel_loss"]) - float(ref["mel_loss"].detach())) < 1e-3
    assert errs["mel"] < 5e-2 and errs["alignment1"] < 5e-2
    # bf16 rounding flips a few ReLU / max-pool decisions (discrete gradient changes), so gradients are judged
    # by direction and L2 error, not max-norm: cosine > 0.98, relative L2 < 0.2 for every parameter tensor
    bad = {}
    for k in grads:
        a, b = grads[k].astype(np.float64).ravel(), gref[k].astype(np.float64).ravel()
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        l2 = float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
        print("%-28s cos=%.5f relL2=%.3e" % (k, cos, l2))
        if not (cos > 0.98 and l2 < 0.2):
            bad[k] = (cos, l2)
    assert not bad, bad


def test_full_config_invariants():
    """BASELINE configs[1] shapes (B=4 to keep it quick): invariants the domain offers at full size —
    alignment rows sum to 1 and are zero beyond source_length, encoder outputs zero beyond length, finite loss."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    ops.set_precision("bf16")
    cfg = ModelConfig()
    eng = Engine(cfg, "cuda", param_seed=0, rng_seed=5)
    batch = synthetic_batch(4, 160, 800, seed=1234)
    b = eng.to_device_batch(batch)
    eng.zero_grad()
    ctx = eng.forward(b, True)
    eng.backward(ctx)
    torch.cuda.synchronize()
    eng.check_clusters(ctx)
    o = eng.outputs(ctx)
    al1 = o["alignment1"].cpu().numpy(); al2 = o["alignment2"].cpu().numpy()
    assert np.allclose(al1.sum(-1), 1.0, atol=1e-4) and np.allclose(al2.sum(-1), 1.0, atol=1e-4)
    for i, L in enumerate(batch["source_length"]):
        assert np.all(al1[i, :, L:] == 0) and np.all(al2[i, :, L:] == 0)
        assert np.all(o["lstm_out"][i, L:].cpu().numpy() == 0)
    assert np.isfinite(float(o["loss"]))
    assert torch.isfinite(eng.grad).all()
    assert float(eng.grad.abs().max()) > 0


def test_training_reduces_loss_on_a_fixed_batch():
    """40 optimiser steps (clip + TF-Adam + Noam schedule, dropout/zoneout on, bf16 benchmark precision) on one fixed
    batch: the loss must fall markedly and stay finite - end-to-end check of forward, backward and optimiser together."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    ops.set_precision("bf16")
    cfg, P = make_params(MEDIUM, seed=4)
    batch = small_batch(cfg, 8, 24, 40, seed=9)
    eng = Engine(cfg, "cuda", params=P, rng_seed=3, lr0=2e-3, decay=False)      # constant rate: no 4000-step warm-up
    b = eng.to_device_batch(batch)
    losses = []
    for _ in range(40):
        ctx = eng.train_step(b)
        losses.append(float(eng.losses[2]))
        eng.optimizer_step()
    eng.check_clusters(ctx)
    print("loss: first %.4f  last %.4f" % (losses[0], losses[-1]))
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.8 * losses[0], losses        # atomics reorder sums run to run: the trajectory is not bit-stable


def test_engine_trains_on_batches_read_from_tfrecord_files(tmp_path):
    """records on disk -> TensorFlow-free reader -> padded batch (a0 contract) -> one train step on the GPU"""
    import copy
    from satt_amd.datasets import ljspeech
    from satt_amd.engine import Engine
    from satt_amd.hparams import hparams as default_hparams
    from satt_amd.utils import tfrecord
    from satt_amd import ops
    cfg, P = make_params(MEDIUM, seed=4)
    h = copy.deepcopy(default_hparams)
    h.parse("dataset=ljspeech.dataset.DatasetSource,batch_size=4,outputs_per_step=%d,num_mels=%d" % (cfg.r, cfg.num_mels))
    h.average_mel_level_db, h.stddev_mel_level_db = [0.0], [1.0]
    g = np.random.default_rng(0)
    src, tgt = [], []
    for i, (L, T) in enumerate([(12, 21), (20, 30), (9, 17), (15, 26)]):
        s = g.integers(1, cfg.num_symbols, L).astype("<i8")
        ps, pt = str(tmp_path / ("u%d.source.tfrecord" % i)), str(tmp_path / ("u%d.target.tfrecord" % i))
        tfrecord.write_records(ps, [tfrecord.make_example({"id": i, "key": b"u%d" % i, "source": s.tobytes(),
                                                           "source_length": L, "text": b"t"})])
        mel = g.normal(0, 1, (T, cfg.num_mels)).astype("<f4")
        tfrecord.write_records(pt, [tfrecord.make_example({"id": i, "key": b"u%d" % i, "mel": mel.tobytes(),
                                                           "mel_width": cfg.num_mels, "target_length": T})])
        src.append(ps); tgt.append(pt)
    batch = next(ljspeech.dataset_factory(src, tgt, h).prepare_and_zip().filter_by_max_output_length().group_by_batch())
    ops.set_precision("f32")
    eng = Engine(cfg, "cuda", params=P, rng_seed=3)
    b = eng.to_device_batch({k: v for k, v in batch.items() if isinstance(v, np.ndarray) and k != "id"})
    ctx = eng.train_step(b)
    eng.check_clusters(ctx)
    assert np.isfinite(float(eng.losses[2])) and bool(torch.isfinite(eng.grad).all())
    assert eng.outputs(ctx)["mel"].shape == (4, int(batch["target_length"].max()), cfg.num_mels)
    # the page-locked ring: batches of a pinned pipeline carry the lease of their slot, the upload leaves a fence, and
    # the ring waits for it before it hands the slot out again; the uploaded values are the slot's values at upload time
    it = iter(ljspeech.dataset_factory(src, tgt, h).prepare_and_zip().repeat().group_by_batch().prefetch(1, pin_memory=True))
    first = next(it)
    ring, slot = first.pinned
    assert torch.as_tensor(first["mel"]).is_pinned() and not ring.fences
    want = first["mel"].copy()
    dev = eng.to_device_batch({k: v for k, v in first.items() if isinstance(v, np.ndarray) and k != "id"}, lease=first.pinned)
    assert slot in ring.fences
    for _ in range(ring.slots + 1):             # draw until the slot has come round: take() must have consumed the fence first
        next(it)
    assert slot not in ring.fences or ring.fences[slot].query()
    torch.cuda.synchronize()
    assert np.array_equal(dev["mel"].cpu().numpy(), want)


def test_unsupported_cluster_shape_falls_back_to_single_workgroup_kernels():
    """Ti > 384 exceeds the single-pass gathers of the attention cluster: the engine must pick the single-workgroup
    kernels by itself and still match the oracle"""
    from satt_amd import ops
    cfg, P = make_params(SMALL, seed=1)
    batch = small_batch(cfg, 2, 390, 8, seed=3)
    ref, col, gref = oracle_run(SMALL, P, batch, True, seed=7)
    eng, out, grads = run_engine(cfg, P, batch, 7, "f32", clusters=True)
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, {}, gref, ["alignment1", "alignment2", "mel", "loss"])
    bad = {k: e for k, e in errs.items() if not (e < 2e-4)}
    assert not bad, bad


@pytest.mark.parametrize("buckets", [3, 2])
def test_dp_bucket_callbacks_leave_the_step_unchanged(buckets):
    """The data-parallel hooks of Engine.train_step on one GPU: the decoder bucket is issued from a side stream as soon
    as its gradients are final, (3 buckets) the encoder's upper half - enc.proj1.W .. enc.sa - in front of the conv-bank
    backward, the rest at the end.  With a stand-in all-reduce (x2 then /2 on the slice, on
    the stream the callback runs on) gradients, loss and the parameter update must match the plain path - i.e. every
    gradient of a bucket is complete (and ordered before the callback) when the callback touches it."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    ops.set_precision("bf16")

    def run(use_cb):
        eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
        eng.dp_buckets = buckets
        b = eng.to_device_batch(synthetic_batch(4, 64, 96, seed=7, min_source_length=30, min_target_steps=20))
        calls = []

        def ar(lo, hi):
            calls.append((lo, hi))
            eng.grad[lo:hi].mul_(2.0); eng.grad[lo:hi].mul_(0.5)
        for _ in range(2):
            eng.train_step(b, allreduce=ar if use_cb else None)
            eng.optimizer_step()
        torch.cuda.synchronize()
        return eng.grad.clone(), eng.flat.clone(), float(eng.losses[2]), calls, eng
    g0, p0, l0, _, _ = run(False)
    g1, p1, l1, calls, eng = run(True)
    want = [(eng.enc_end, eng.nparam), (eng.enc_mid, eng.enc_end), (0, eng.enc_mid)] if buckets == 3 else \
        [(eng.enc_end, eng.nparam), (0, eng.enc_end)]
    assert calls[-len(want):] == want                                       # decoder bucket first, the conv bank's last
    assert abs(l0 - l1) < 1e-4
    assert float((g0 - g1).abs().max()) < 1e-3 * float(g0.abs().max())     # atomics order is the only difference
    assert float((p0 - p1).abs().max()) < 1e-5


def test_full_size_bf16_tracks_f32():
    """BASELINE configs[1] shapes (Ti=160, Tm=800; B=4): the benchmark precision (bf16 MFMA operands, bf16 keys/values in
    LDS) against the exact-fp32-GEMM mode of the same engine on the same batch and masks.  This is a CONSISTENCY check of
    the two modes, not a parity claim: both consume the same bf16 recurrent weights, so the rounding of those weights
    cancels here.  The parity claim at this size - unrounded weights against the float64 oracle, both modes - is
    test_full_size_unrounded_weights_vs_oracle below."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    batch = synthetic_batch(4, 160, 800, seed=77)
    res = {}
    for prec in ("f32", "bf16"):
        ops.set_precision(prec)
        eng = Engine(ModelConfig(), "cuda", param_seed=3, rng_seed=5)
        b = eng.to_device_batch(batch)
        eng.zero_grad()
        ctx = eng.forward(b, True)
        eng.backward(ctx)
        torch.cuda.synchronize()
        eng.check_clusters(ctx)
        o = eng.outputs(ctx)
        res[prec] = dict(mel_loss=float(o["mel_loss"]), loss=float(o["loss"]), al1=o["alignment1"].cpu().numpy(),
                         grad=eng.grad.detach().cpu().numpy().astype(np.float64))
    ops.set_precision("bf16")
    print(res["f32"]["mel_loss"], res["bf16"]["mel_loss"])
    assert abs(res["f32"]["mel_loss"] - res["bf16"]["mel_loss"]) < 1e-3
    assert abs(res["f32"]["loss"] - res["bf16"]["loss"]) < 2e-3
    assert np.abs(res["f32"]["al1"] - res["bf16"]["al1"]).max() < 5e-2
    a, b_ = res["bf16"]["grad"], res["f32"]["grad"]
    cos = float(a @ b_ / (np.linalg.norm(a) * np.linalg.norm(b_) + 1e-30))
    print("gradient cosine", cos)
    assert cos > 0.98


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_single_launch_attention_equals_chunked_launches(prec):
    """The attention kernels span all pipeline chunks in one launch and hand chunks over through per-chunk counters
    (hipStreamWaitValue32 / in-kernel waits) instead of kernel boundaries.  Samples with short texts run ahead by whole
    chunks, so the hand-off must be per chunk AND per workgroup count: with full sequence lengths and ragged source
    lengths the result has to equal the one-launch-per-chunk schedule (which needs no such signalling)."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    # (bf16, r5: the folded forward kernel carries its forward variable un-normalised inside a launch and restarts a chunked launch
    #  from the saved NORMALISED rows - the two schedules differ by roundings only)
    ops.set_precision(prec)
    batch = synthetic_batch(8 if prec == "bf16" else 4, 160, 800, seed=77)
    res = {}
    for single in (False, True):
        eng = Engine(ModelConfig(), "cuda", param_seed=3, rng_seed=5)
        eng.single_launch_attention = single
        b = eng.to_device_batch(batch)
        for _ in range(2):                       # the second pass reuses buffers that hold plausible stale values
            eng.zero_grad()
            ctx = eng.forward(b, True)
            eng.backward(ctx)
            torch.cuda.synchronize()
            eng.check_clusters(ctx)
        res[single] = (float(eng.losses[2]), ctx["h1"].clone(), eng.grad.clone())
    ops.set_precision("bf16")
    tol = 1.0          # (measured: f32 1.2e-7 / 3.7e-7 / 1.6e-8, bf16 2.4e-7 / 3.8e-5 / 1.4e-5)
    print("single vs chunked launches (%s): |d loss| %.2e, max |d h1| %.2e, max |d grad| / max |grad| %.2e" % (
        prec, abs(res[True][0] - res[False][0]), float((res[True][1] - res[False][1]).abs().max()),
        float((res[True][2] - res[False][2]).abs().max()) / float(res[False][2].abs().max())))
    assert abs(res[True][0] - res[False][0]) < 1e-5 * tol
    assert float((res[True][1] - res[False][1]).abs().max()) < 1e-4 * tol
    gd = float((res[True][2] - res[False][2]).abs().max())
    assert gd < 1e-3 * tol * float(res[False][2].abs().max()), gd


def test_full_size_unrounded_weights_vs_oracle():
    """BASELINE configs[1] at FULL length (Ti=160, Tm=800) with B = 8 - the batch size of configs[0], a multiple of 8 so
    the same-XCD plain-store exchange runs - and fp32 master weights straight from init_params: NO slice is made
    bf16-representable beforehand, so the bf16 rounding of the recurrent weights (consumed as bf16 by the persistent
    kernels in BOTH precisions) compounds over the 400 decoder steps exactly as in training.  Judge: the float64 oracle
    on the same batch and masks.  Bar (BASELINE.json): |mel_loss - oracle| < 1e-3; alignment error and per-tensor
    gradient cosines are reported and bounded."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig, init_params
    from satt_amd.datasets.synthetic import synthetic_batch
    torch.set_num_threads(min(16, torch.get_num_threads()))      # tiny per-step ops: more threads only add overhead
    B = 8
    cfg = ModelConfig()
    P = init_params(cfg, 3)
    batch = synthetic_batch(B, 160, 800, seed=77)
    ref, col, gref = oracle_run(dict(), P, batch, True, seed=5)
    ref_mel_loss = float(ref["mel_loss"].detach())
    ref_al1, ref_al2 = ref["alignment1"].detach().numpy(), ref["alignment2"].detach().numpy()
    rows = {}
    for prec in ("f32", "bf16"):
        eng, out, grads = run_engine(cfg, P, batch, 5, prec)
        assert_same_xcd_fast_path(eng, B)
        d_mel = abs(float(out["mel_loss"]) - ref_mel_loss)
        d_loss = abs(float(out["loss"]) - float(ref["loss"].detach()))
        e_al1 = float(np.abs(out["alignment1"] - ref_al1).max())
        e_al2 = float(np.abs(out["alignment2"] - ref_al2).max())
        e_mel = float(np.abs(out["mel"] - ref["mel"].detach().numpy()).max())
        cosines = {}
        for k in grads:
            a, b = grads[k].astype(np.float64).ravel(), gref[k].astype(np.float64).ravel()
            cosines[k] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        ga = np.concatenate([grads[k].astype(np.float64).ravel() for k in grads])
        gb = np.concatenate([gref[k].astype(np.float64).ravel() for k in grads])
        cos_all = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
        worst = min(cosines, key=cosines.get)
        print("[full size, unrounded weights] %s: |d mel_loss|=%.3e |d loss|=%.3e  max|d mel|=%.3e  "
              "max|d align1|=%.3e max|d align2|=%.3e  grad cos(all)=%.6f  worst tensor %s cos=%.5f"
              % (prec, d_mel, d_loss, e_mel, e_al1, e_al2, cos_all, worst, cosines[worst]))
        rows[prec] = (d_mel, d_loss, e_al1, e_al2, cos_all, cosines[worst])
    ops.set_precision("bf16")
    for prec, (d_mel, d_loss, e_al1, e_al2, cos_all, cos_worst) in rows.items():
        assert d_mel < 1e-3, (prec, d_mel)
        assert d_loss < 2e-3, (prec, d_loss)
        # measured on MI355X (round 2): f32 |d mel_loss| 1.3e-6, align 5.9e-4, cos 0.999999, worst tensor 1.00000;
        # bf16 3.3e-6, 2.7e-3, 0.99998, worst tensor 0.99707.  Bars = about 3x the measured distance (r3; they were 10x looser)
        if prec == "f32":
            assert d_mel < 1e-4 and e_al1 < 3e-3 and e_al2 < 3e-3, (prec, d_mel, e_al1, e_al2)
            assert cos_all > 0.99999 and cos_worst > 0.9999, (prec, cos_all, cos_worst)
        else:
            assert e_al1 < 1e-2 and e_al2 < 1e-2, (prec, e_al1, e_al2)
            assert cos_all > 0.9999 and cos_worst > 0.995, (prec, cos_all, cos_worst)


@pytest.mark.parametrize("attention,cumulative", [("location_sensitive", False), ("location_sensitive", True), ("forward", True)])
@pytest.mark.parametrize("cfg_kw,B,Ti,Tm", [(MEDIUM, 5, 37, 46), (dict(), 8, 21, 24)])
def test_f32_parity_attention_options(cfg_kw, B, Ti, Tm, attention, cumulative):
    """hparams `attention=location_sensitive` (softmax alignments, no alpha recursion; modules/attentions.py:35-42) and
    `cumulative_weights=True` (modules/forward_attention.py:118-119) on the cluster kernels: every forward tensor and all
    parameter gradients against the float64 oracle, dropout / zoneout on."""
    kw = dict(cfg_kw, attention=attention, cumulative_weights=cumulative)
    cfg, P = make_params(kw, seed=1)
    batch = small_batch(cfg, B, Ti, Tm, seed=3)
    g = np.random.default_rng(0)
    Td = Tm // cfg.r
    dal = (g.normal(0, 0.05, (B, Td, Ti)), g.normal(0, 0.05, (B, Td, Ti)))
    ref, col, gref = oracle_run(kw, P, batch, True, seed=7, dalign=dal)
    eng, out, grads = run_engine(cfg, P, batch, 7, "f32", dalign=dal, clusters=True)
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref,
                  ["lstm_out", "sa_out", "alignment1", "alignment2", "dec_out", "mel", "stop", "loss"])
    bad = {k: e for k, e in errs.items() if not (e < 2e-4)}
    assert not bad, bad
    if attention == "location_sensitive":      # the returned alignments ARE the softmax probabilities
        assert np.allclose(out["alignment1"], eng.last_ctx["a1"].cpu().numpy(), atol=1e-6)


BASELINE_DIMS = dict(sa_units=0, att2_units=0, dec_sa_units=0, att1_units=256)     # examples/ljspeech/tacotron.json at production size
MODELS = {"self-attention": dict(), "baseline": BASELINE_DIMS}


def baseline_kw(kw):
    """ExtendedTacotronV1Model (reference models/models.py:20-226, examples/ljspeech/tacotron.json): ZoneoutEncoderV1 +
    ExtendedDecoder v2 = one attention source, no self-attention blocks; attention num_units = attention_out_units"""
    out = dict(kw, sa_units=0, att2_units=0, dec_sa_units=0)
    out["att1_units"] = kw.get("att_rnn_units", 256) if kw else 256
    return out


@pytest.mark.parametrize("clusters", [True, False])
@pytest.mark.parametrize("cfg_kw,B,Ti,Tm", [(SMALL, 3, 9, 12), (MEDIUM, 5, 37, 46), (MEDIUM, 8, 29, 34), (dict(), 8, 21, 24)])
def test_f32_parity_baseline_tacotron(cfg_kw, B, Ti, Tm, clusters):
    """the single-source variant through the same kernels (U2 = V2 = 0, second-mechanism pointers NULL) against the
    float64 oracle: outputs, alignments and every parameter gradient"""
    kw = baseline_kw(cfg_kw)
    if not clusters and not cfg_kw:
        pytest.skip("production dimensions run on the cluster kernels")
    cfg, P = make_params(kw, seed=11)
    assert not cfg.dual and "dec.att2.v" not in P and "enc.sa.kvq.W" not in P and "dec.sa.kvq.W" not in P
    batch = small_batch(cfg, B, Ti, Tm, seed=12)
    g = np.random.default_rng(2)
    Td = Tm // cfg.r
    dal = (g.normal(0, 0.05, (B, Td, Ti)), np.zeros((B, Td, Ti)))
    ref, col, gref = oracle_run(kw, P, batch, True, seed=13, dalign=dal)
    eng, out, grads = run_engine(cfg, P, batch, 13, "f32", dalign=dal, clusters=clusters)
    if clusters and B % 8 == 0:
        assert_same_xcd_fast_path(eng, B)
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref,
                  ["lstm_out", "alignment1", "dec_out", "mel", "stop", "loss", "mel_loss", "done_loss"])
    bad = {k: e for k, e in errs.items() if not (e < 2e-4)}
    assert not bad, bad


@pytest.mark.parametrize("B,Ti,Tm", [(8, 21, 24), (3, 45, 60)])
def test_bf16_parity_baseline_tacotron_production_dims(B, Ti, Tm):
    """the baseline model at its production dimensions in bf16: the cluster kernels' second compile-time specialisation
    (SpecDimsOf<2>: one source, 256 attention units) against the float64 oracle, same bars as test_bf16_parity"""
    kw = baseline_kw(dict())
    cfg, P = make_params(kw, seed=31)
    batch = small_batch(cfg, B, Ti, Tm, seed=32)
    ref, col, gref = oracle_run(kw, P, batch, True, seed=33)
    eng, out, grads = run_engine(cfg, P, batch, 33, "bf16", clusters=True)
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref,
                  ["lstm_out", "alignment1", "dec_out", "mel", "stop", "loss", "mel_loss", "done_loss"])
    assert abs(float(out["mel_loss"]) - float(ref["mel_loss"].detach())) < 1e-3
    assert errs["mel"] < 5e-2 and errs["alignment1"] < 5e-2
    bad = {}
    for k in grads:
        a, b = grads[k].astype(np.float64).ravel(), gref[k].astype(np.float64).ravel()
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        l2 = float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
        if not (cos > 0.98 and l2 < 0.2):
            bad[k] = (cos, l2)
    assert not bad, bad


@pytest.mark.parametrize("cfg_kw,B", [(MEDIUM, 4), (dict(), 8)])
def test_f32_parity_baseline_tacotron_multi_speaker(cfg_kw, B):
    """examples/vctk/tacotron.json: the baseline model with the speaker embedding fed to the decoder pre-net
    (reference models/models.py:40-43,48,57: speaker_embed -> ExtendedDecoder -> MultiSpeakerPreNet, module.py:573-577)"""
    kw = dict(baseline_kw(cfg_kw), num_speakers=7, speaker_dim=16, speaker_offset=225)
    cfg, P = make_params(kw, seed=21)
    batch = small_batch(cfg, B, 21, 26, seed=22)
    batch["speaker_id"] = (np.random.default_rng(3).integers(0, 7, B) + 225).astype(np.int64)
    ref, col, gref = oracle_run(kw, P, batch, True, seed=23)
    eng, out, grads = run_engine(cfg, P, batch, 23, "f32")
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref, ["mel", "stop", "alignment1", "loss"])
    bad = {k: e for k, e in errs.items() if not (e < 2e-4)}
    assert not bad, bad
    assert float(np.abs(grads["speaker_embedding"]).max()) > 0


@pytest.mark.parametrize("cfg_kw,B,Ti,Tm,cum", [(MEDIUM, 5, 37, 46, False), (MEDIUM, 8, 29, 34, True), (dict(), 8, 21, 24, False)])
def test_f32_parity_transition_agent(cfg_kw, B, Ti, Tm, cum):
    """use_forward_attention_transition_agent (reference modules/forward_attention.py:80-86,111-116): the transition
    probability of the forward recursion is predicted per step from [context | processed query]; outputs, alignments and
    every gradient (incl. the agent's Dense) against the float64 oracle"""
    kw = dict(cfg_kw, transition_agent=True, cumulative_weights=cum)
    cfg, P = make_params(kw, seed=31)
    P["dec.att1.Wa"] = (3.0 * P["dec.att1.Wa"]).astype(np.float32)        # move u well away from 0.5
    # (seed 32 puts one highway-0 ReLU of the B = 8 batch within fp32 rounding of its kink - with or without the agent - and
    # the flipped unit shows up as 1e-3 in the gradients below it; see test_f32_parity_production_dims on such ties)
    batch = small_batch(cfg, B, Ti, Tm, seed=77)
    g = np.random.default_rng(4)
    Td = Tm // cfg.r
    dal = (g.normal(0, 0.05, (B, Td, Ti)), g.normal(0, 0.05, (B, Td, Ti)))
    ref, col, gref = oracle_run(kw, P, batch, True, seed=33, dalign=dal)
    eng, out, grads = run_engine(cfg, P, batch, 33, "f32", dalign=dal, clusters=True)
    us = eng.last_ctx["ustate"][:, 1:].float().cpu().numpy()
    assert np.abs(us - 0.5).max() > 0.02, "the agent never moved the transition probability"
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref,
                  ["lstm_out", "alignment1", "alignment2", "dec_out", "mel", "stop", "loss"])
    bad = {k: e for k, e in errs.items() if not (e < 2e-4)}
    assert not bad, bad
    # chunked launches carry d u between chunks: identical gradients
    eng2, out2, grads2 = run_engine_chunked(cfg, P, batch, 33, dal)
    for k in ("dec.att1.Wa", "dec.att1.ba", "dec.att_lstm.W", "enc.lstm_fw.W"):
        assert rel_err(grads2[k], grads[k]) < 1e-5, k


@pytest.mark.parametrize("cfg_kw,hops,B,Ti,Tm", [(SMALL, (2, 3), 3, 9, 12), (MEDIUM, (3, 2), 5, 37, 46), (dict(), (2, 2), 2, 21, 24)])
def test_f32_parity_multi_hop_transformers(cfg_kw, hops, B, Ti, Tm):
    """self_attention_num_hop / decoder_self_attention_num_hop > 1 (reference modules/module.py:411-419 + :433-439, :707-715 +
    :753-757): stacked SelfAttentionTransformer blocks with their own weights, in the encoder's second source and behind the
    decoder RNN; outputs and every parameter gradient against the float64 oracle (dropout on, one mask stream per hop)."""
    kw = dict(cfg_kw, sa_num_hop=hops[0], dec_sa_num_hop=hops[1])
    cfg, P = make_params(kw, seed=21)
    batch = small_batch(cfg, B, Ti, Tm, seed=23)
    ref, col, gref = oracle_run(kw, P, batch, True, seed=29)
    eng, out, grads = run_engine(cfg, P, batch, 29, "f32")
    errs = report(out, {**ref, "dec_out": col["dec_out"]}, grads, gref,
                  ["lstm_out", "sa_out", "alignment1", "alignment2", "dec_out", "mel", "stop", "loss"])
    bad = {k: e for k, e in errs.items() if not (e < 2e-4)}
    assert not bad, bad
    for h in range(1, hops[0]):
        assert float(np.abs(grads["enc.sa.h%d.kvq.W" % h]).max()) > 0
        e = rel_err(out["enc_alignments"][h], col["enc_alignments"][h].detach().numpy())
        assert e < 2e-4, ("enc_alignments", h, e)
    for h in range(1, hops[1]):
        assert float(np.abs(grads["dec.sa.h%d.t.W" % h]).max()) > 0
    # benchmark precision: same step within the bf16 bars of the single-hop tests
    eng2, out2, grads2 = run_engine(cfg, P, batch, 29, "bf16")
    rml = float(ref["mel_loss"].detach())
    assert abs(float(out2["mel_loss"]) - rml) < 1e-3 * max(1.0, rml)
    a = np.concatenate([grads2[k].ravel() for k in sorted(grads2)]); b = np.concatenate([gref[k].ravel() for k in sorted(grads2)])
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    print("bf16 grad cosine", cos)
    assert cos > 0.995, cos


@pytest.mark.parametrize("model", ["self-attention", "baseline"])
@pytest.mark.parametrize("B,Ti,Tm", [(2, 21, 24), (8, 160, 120), (4, 97, 64), (3, 160, 200)])
def test_saved_attention_factors_equal_the_recomputation(B, Ti, Tm, model):
    """satt_attn_rnn_params.saf: the folded forward kernel saves r (1 - r) of the energy nonlinearity per (step, memory row, unit)
    as fp16 and the backward kernel reads it instead of recomputing it (csrc/attn_cluster.hip, SAF).  Same step with and without:
    identical forward, gradients equal to fp16 rounding of one factor (2^-11 per element, uncorrelated)."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    ops.set_precision("bf16")
    cfg = ModelConfig(**MODELS[model])
    batch = synthetic_batch(B, Ti, Tm, seed=77)
    res = {}
    for on in (True, False):
        eng = Engine(cfg, "cuda", param_seed=5, rng_seed=9)
        eng.save_attention_factors = on
        b = eng.to_device_batch(batch)
        for _ in range(2):                              # the second pass runs on recycled buffers
            eng.zero_grad()
            ctx = eng.forward(b, True)
            eng.backward(ctx)
        torch.cuda.synchronize()
        eng.check_clusters(ctx)
        assert ("saf" in ctx) == on
        res[on] = (float(eng.losses[2]), eng.grad.detach().double().cpu().numpy(), {k: v.detach().double().cpu().numpy() for k, v in eng.G.items()})
    assert abs(res[True][0] - res[False][0]) < 2e-6                       # the forward pass is untouched (block sums of the loss: atomics)
    a, bb = res[True][1], res[False][1]
    cos = float(a @ bb / (np.linalg.norm(a) * np.linalg.norm(bb)))
    # per tensor: max |difference| against the tensor's own scale plus an absolute floor - the location layer's gradient
    # (dec.att1.U) is a sum of terms that cancel to ~1e-6 of their size on this random batch: its last digits depend on the
    # order of the partial sums (pieces of the deferred gradients), 5e-8 absolute
    worst, wk = max((float(np.abs(res[True][2][k] - res[False][2][k]).max() / (np.abs(res[False][2][k]).max() + 2e-6)), k) for k in res[True][2])
    print("saved factors vs recomputation: cosine %.8f, worst tensor max-rel %.2e (%s)" % (cos, worst, wk))
    assert cos > 0.99999 and worst < 2e-2, (cos, worst, wk)


@pytest.mark.parametrize("B,Ti,Tm,reps,model", [(3, 160, 200, 5, "self-attention"), (32, 160, 800, 2, "self-attention"),
                                                 (32, 80, 500, 2, "self-attention"), (32, 160, 800, 2, "baseline"),
                                                 (5, 97, 330, 3, "baseline")])
def test_identical_steps_give_identical_gradients(B, Ti, Tm, reps, model):
    """Two engines, same seeds, same batch, the overlapped schedule of the benchmark (side streams, single-launch attention, saved
    factors): every gradient tensor agrees to 1e-4 of its own largest element.  Float atomics reorder sums (1e-6), nothing else may
    differ - found in round 3: one packed-math form of the deferred location-layer gradient came out 1e-8 wrong on 16 elements
    (3 % of that near-zero tensor) only when it ran beside the recurrent kernels (tools/probes/grad_diff_map.py)."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    ops.set_precision("bf16")
    batch = synthetic_batch(B, Ti, Tm, seed=77)

    def run():
        eng = Engine(ModelConfig(**MODELS[model]), "cuda", param_seed=5, rng_seed=9)
        b = eng.to_device_batch(batch)
        eng.zero_grad(); ctx = eng.forward(b, True); eng.backward(ctx)
        torch.cuda.synchronize(); eng.check_clusters(ctx)
        assert ctx["single_launch_bwd"] and "saf" in ctx
        return {k: v.detach().double().cpu().numpy() for k, v in eng.G.items()}
    ref = run()
    for rep in range(reps):
        g = run()
        bad = {k: float(np.abs(g[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)) for k in ref}
        bad = {k: e for k, e in bad.items() if e > 1e-4}
        assert not bad, (rep, bad)


@pytest.mark.parametrize("model", ["self-attention", "baseline"])
def test_fused_lstm_input_projection_equals_the_separate_gemm(model):
    """csrc/lstm_cluster.hip, fused input projection (satt_lstm_cluster_fwd_x): the short forward chunks form x W_in + b inside the
    LSTM cluster launch - same operand rounding, same accumulation order as the GEMM: the saved gates and outputs of both decoder
    LSTMs are BIT-identical with the switch on (every chunk fused here: 64 steps) and off."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    ops.set_precision("bf16")
    cfg = ModelConfig(**MODELS[model])
    batch = synthetic_batch(8, 64, 200, seed=21)
    got = {}
    for steps in (0, 64):
        eng = Engine(cfg, "cuda", param_seed=5, rng_seed=9)
        eng.fuse_xg_steps = steps
        ctx = eng.forward(eng.to_device_batch(batch), True)
        torch.cuda.synchronize(); eng.check_clusters(ctx)
        assert ctx["chunks"] > 1
        got[steps] = [ctx["h1"].clone()] + [t.clone() for t in ctx["l1"]] + [t.clone() for t in ctx["l2"]]
    for a, b in zip(got[0], got[64]):
        assert torch.equal(a, b)


def test_one_engine_steps_through_growing_batch_shapes():
    """Buffers an engine keeps across steps must follow the problem size: the float64 slots of the deferred attention gradients are
    one per workgroup of a grid that grows with B * Ti (r4: a buffer sized by the first batch was overrun by the next, larger one -
    a crash in the drivers' tests, whose batches differ in shape).  Small batch first, then a larger one on the SAME engine: the
    gradients equal those of a fresh engine on the larger batch."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    ops.set_precision("bf16")
    cfg = ModelConfig(**MODELS["self-attention"])
    small, large = synthetic_batch(2, 24, 40, seed=3), synthetic_batch(8, 160, 120, seed=4)

    def step(eng, batch):
        b = eng.to_device_batch(batch)
        eng.zero_grad(); ctx = eng.forward(b, True); eng.backward(ctx)
        torch.cuda.synchronize(); eng.check_clusters(ctx)
        assert "saf" in ctx
        return {k: v.detach().double().cpu().numpy() for k, v in eng.G.items()}
    eng = Engine(cfg, "cuda", param_seed=5, rng_seed=9)
    step(eng, small)
    got = step(eng, large)
    ref = step(Engine(cfg, "cuda", param_seed=5, rng_seed=9), large)
    bad = {k: float(np.abs(got[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)) for k in ref}
    bad = {k: e for k, e in bad.items() if e > 1e-4}
    assert not bad, bad


def run_engine_chunked(cfg, P, batch, seed, dalign):
    from satt_amd import ops
    from satt_amd.engine import Engine
    ops.set_precision("f32")
    eng = Engine(cfg, "cuda", params=P, rng_seed=seed)
    eng.single_launch_attention = False
    b = eng.to_device_batch(batch)
    eng.zero_grad()
    ctx = eng.forward(b, training=True)
    ctx["dalign1"] = torch.as_tensor(dalign[0], dtype=torch.float32, device="cuda").contiguous()
    ctx["dalign2"] = torch.as_tensor(dalign[1], dtype=torch.float32, device="cuda").contiguous()
    eng.backward(ctx)
    torch.cuda.synchronize()
    eng.check_clusters(ctx)
    return eng, None, {k: v.detach().cpu().numpy() for k, v in eng.G.items()}


def test_f32_parity_baseline_l2_regularization():
    """use_l2_regularization (reference modules/regularizers.py:11-18; read by the baseline model_fn only, models/models.py:
    109-114): loss += weight * sum ||W||^2 / 2 over the non-blacklisted kernels, gradients += weight * W"""
    kw = dict(baseline_kw(MEDIUM), l2_weight=3e-3, transition_agent=True)
    cfg, P = make_params(kw, seed=41)
    batch = small_batch(cfg, 5, 37, 46, seed=77)
    ref, col, gref = oracle_run(kw, P, batch, True, seed=43)
    eng, out, grads = run_engine(cfg, P, batch, 43, "f32")
    assert float(ref["regularization_loss"].detach()) > 1e-2 * float(ref["mel_loss"].detach())    # a visible share of the loss
    errs = report(out, ref, grads, gref, ["mel", "loss", "regularization_loss", "mel_loss"])
    bad = {k: e for k, e in errs.items() if not (e < 2e-4)}
    assert not bad, bad
    plain = oracle_run(dict(kw, l2_weight=0.0), P, batch, True, seed=43)[2]
    assert rel_err(plain["enc.bank3.W"], gref["enc.bank3.W"]) > 1e-3 and np.array_equal(plain["dec.lstm1.W"], gref["dec.lstm1.W"])


@pytest.mark.parametrize("cfg_kw,B,Ti,Tm,clusters,decay", [(SMALL, 3, 9, 12, False, False), (MEDIUM, 8, 24, 40, True, False),
                                                           (MEDIUM, 5, 37, 46, True, True)])
def test_training_trajectory_matches_oracle(cfg_kw, B, Ti, Tm, clusters, decay):
    """Five optimisation steps on five different batches against the float64 oracle run as a training loop of its own
    (forward + autograd + tf.clip_by_global_norm + TF-Adam + the reference's rate schedule, models/models.py:485-498,
    594-598; dropout / zoneout seed advancing by one per step): the loss and the global gradient norm of every step and the
    accumulated parameter update.  After the first update the recurrent weights are no longer bf16-representable, so the
    kernels' bf16 shadows differ from the oracle's weights by their rounding: the bar is that rounding, not fp32."""
    from oracle import torch_ref
    from satt_amd import ops
    from satt_amd.engine import Engine
    from common import oracle_cfg
    ops.set_precision("f32")
    cfg, P0 = make_params(cfg_kw, seed=51)
    lr0, sf, seed0, K = 2e-3, (700.0 if decay else 1.0), 61, 5
    eng = Engine(cfg, "cuda", params=P0, rng_seed=seed0, lr0=lr0, decay=decay, step_factor=sf)
    eng.use_clusters = clusters
    ocfg = oracle_cfg(cfg_kw)
    Po = torch_ref.to_torch(P0, torch.float64)
    m = {k: torch.zeros_like(v) for k, v in Po.items()}; v = {k: torch.zeros_like(x) for k, x in Po.items()}
    rows = []
    for t in range(1, K + 1):
        batch = small_batch(cfg, B, Ti, Tm, seed=80 + t)
        # oracle step
        Pt = {k: x.clone().requires_grad_(True) for k, x in Po.items()}
        out = torch_ref.forward(Pt, torch_ref.batch_to_torch(batch), ocfg, True, seed0 + t - 1)
        gl = torch.autograd.grad(out["loss"], list(Pt.values()), allow_unused=True)
        g = {k: (x if x is not None else torch.zeros_like(Po[k])) for k, x in zip(Pt.keys(), gl)}
        lr = torch_ref.learning_rate(lr0, t - 1, sf) if decay else lr0
        gn = torch_ref.clip_and_adam(Po, g, m, v, t, lr)
        # engine step
        ctx = eng.train_step(eng.to_device_batch(batch))
        assert abs(eng.learning_rate() - lr) < 1e-9 * lr0 + 1e-6 * lr
        eng.optimizer_step()
        torch.cuda.synchronize()
        eng.check_clusters(ctx)
        le, lo = float(eng.losses[2]), float(out["loss"].detach())
        gne = float(eng.opt_state[1])
        rows.append((t, lo, le, gn, gne, lr))
        print("step %d  loss oracle %.6f engine %.6f (rel %.2e)  |g| oracle %.5f engine %.5f (rel %.2e)  lr %.3e"
              % (t, lo, le, abs(le - lo) / lo, gn, gne, abs(gne - gn) / gn, lr))
    for t, lo, le, gn, gne, lr in rows:
        assert abs(le - lo) < 2e-3 * lo and abs(gne - gn) < 1e-2 * gn, rows
    assert rows[0][2] != rows[-1][2]
    # accumulated update of every tensor: direction and size (Adam's m / sqrt(v) turns a gradient of rounding-noise size
    # into a full-size step of either sign, so single elements are not comparable - the update vectors are)
    worst = (1.0, None); moved = 0.0
    for k in Po:
        d_o = (Po[k].numpy() - np.asarray(P0[k], dtype=np.float64)).ravel()
        d_e = (eng.P[k].detach().cpu().numpy().astype(np.float64) - np.asarray(P0[k], dtype=np.float64)).ravel()
        no, ne = np.linalg.norm(d_o), np.linalg.norm(d_e)
        moved = max(moved, no)
        if no == 0:
            assert ne == 0, k
            continue
        cos = float(d_o @ d_e / (no * ne))
        print("%-28s |dP| oracle %.4e engine %.4e  cos %.6f" % (k, no, ne, cos))
        assert abs(ne - no) < 0.05 * no, (k, no, ne)
        if cos < worst[0]:
            worst = (cos, k)
    assert moved > 10 * lr0                      # the test moved the parameters by many times the rate
    assert worst[0] > 0.995, worst


def test_cluster_timeout_flag_is_sticky_and_guards_the_update():
    """a hand-off timeout in ANY step must neither reach Adam nor go unnoticed until the next host check.  The
    error word of a cluster workspace lives in its 64-byte tail, which launches never clear (only the owner zeroes it): set
    it by hand after step 1, run more steps - the parameters must stay put and check_clusters() must raise steps later."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    ops.set_precision("bf16")
    cfg, P = make_params(MEDIUM, seed=4)
    batch = small_batch(cfg, 8, 24, 40, seed=9)
    eng = Engine(cfg, "cuda", params=P, rng_seed=3, lr0=2e-3, decay=False)
    b = eng.to_device_batch(batch)
    ctx = eng.train_step(b); eng.optimizer_step()
    eng.check_clusters(ctx)
    p1 = eng.flat.clone()
    ctx = eng.train_step(b); eng.optimizer_step()
    assert not torch.equal(eng.flat, p1)                   # healthy steps update
    Ca, aws = ctx["att_cluster"]
    assert Ca > 0 and aws is eng._ws_last["attn"]
    aws[-64:].view(torch.int32)[0] = 1                     # what a bounded spin writes when it gives up (cluster_xchg.h)
    p2 = eng.flat.clone()
    for _ in range(3):                                     # the flag survives the launches of later steps ...
        ctx = eng.train_step(b); eng.optimizer_step()
    torch.cuda.synchronize()
    assert torch.equal(eng.flat, p2)                       # ... every one of which skipped its update on the device
    with pytest.raises(RuntimeError):
        eng.check_clusters(ctx)
    # the training driver's recovery: event-ordered chunk schedule from here on, error words cleared, training continues
    assert eng.single_launch_attention and eng.recover_from_handoff_timeout()
    assert not eng.single_launch_attention and int(aws[-64:].view(torch.int32)[0]) == 0
    ctx = eng.train_step(b); eng.optimizer_step()
    eng.check_clusters(ctx)
    assert not torch.equal(eng.flat, p2) and not ctx["single_launch_fwd"] and not ctx["single_launch_bwd"]
    assert eng.recover_from_handoff_timeout() is False      # nothing further to fall back to


@pytest.mark.parametrize("model", ["self-attention", "baseline"])
@pytest.mark.parametrize("B,Ti,Tm", [(2, 21, 24), (8, 160, 120), (4, 97, 64)])
def test_folded_context_equals_the_unfolded_kernel(B, Ti, Tm, model):
    """csrc/attn_cluster.hip FOLD: gates += ctx1 Wc1 evaluated as alpha (values1 Wc1) inside the recurrent product, ctx1 itself
    formed by a GEMM outside the kernel - against the unfolded kernel of the same precision on the same batch and masks.  The
    two differ by one bf16 rounding of values1 Wc1 (the unfolded form rounds Wc1 and keeps ctx1 exact), so outputs agree to
    bf16-weight-rounding level; the backward pass is the same code in both runs."""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    ops.set_precision("bf16")
    batch = synthetic_batch(B, Ti, Tm, seed=31, min_source_length=max(2, Ti // 2), min_target_steps=max(2, Tm // 4))
    res = {}
    for fold in (False, True):
        eng = Engine(ModelConfig(**MODELS[model]), "cuda", param_seed=3, rng_seed=5)
        eng.fold_context = fold
        b = eng.to_device_batch(batch)
        eng.zero_grad()
        ctx = eng.forward(b, True)
        eng.backward(ctx)
        torch.cuda.synchronize()
        eng.check_clusters(ctx)
        assert ("saf" in ctx) == fold          # the folded launch is the one that saves the factors
        o = eng.outputs(ctx)
        res[fold] = dict(loss=float(o["loss"]), al1=o["alignment1"].cpu().numpy(), att=ctx["att_out"].cpu().numpy(),
                         mel=o["mel"].cpu().numpy(), grad=eng.grad.detach().cpu().numpy().astype(np.float64))
    a, f = res[False], res[True]
    A = ModelConfig().att_rnn_units
    e_ctx = float(np.abs(a["att"][:, A:] - f["att"][:, A:]).max() / (np.abs(a["att"][:, A:]).max() + 1e-12))
    e_h = float(np.abs(a["att"][:, :A] - f["att"][:, :A]).max())
    e_al = float(np.abs(a["al1"] - f["al1"]).max())
    cos = float(a["grad"] @ f["grad"] / (np.linalg.norm(a["grad"]) * np.linalg.norm(f["grad"]) + 1e-30))
    print("fold vs unfolded: |d loss|=%.3e ctx rel=%.3e max|d h|=%.3e max|d align1|=%.3e grad cos=%.6f"
          % (abs(a["loss"] - f["loss"]), e_ctx, e_h, e_al, cos))
    # measured (r3): |d loss| 7e-6, ctx 3e-6, h 3.8e-4, alignment 1.7e-6, cosine 1.000000 - bars about 4x that
    assert abs(a["loss"] - f["loss"]) < 5e-5
    assert e_ctx < 2e-5 and e_h < 1.5e-3 and e_al < 1e-5
    assert cos > 0.999999
