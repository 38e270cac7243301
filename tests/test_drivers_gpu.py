"""train.py / predict_mel.py with the reference's command lines on a tiny on-disk corpus (LJSpeech config)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_then_predict(tmp_path):
    sys.path.insert(0, ROOT)
    import satt_amd  # noqa: F401
    from satt_amd.utils import tfrecord
    g = np.random.default_rng(0)
    data, lists, ckpt, out = tmp_path / "data", tmp_path / "lists", tmp_path / "ckpt", tmp_path / "out"
    for d in (data, lists, ckpt, out):
        d.mkdir()
    keys = ["LJ%03d" % i for i in range(6)]
    raw_len = {}
    for i, k in enumerate(keys):
        L, T = int(g.integers(8, 16)), int(g.integers(20, 40))
        raw_len[k] = T
        s = np.concatenate([[0], g.integers(1, 60, L - 2), [0]]).astype("<i8")
        tfrecord.write_records(str(data / (k + ".source.tfrecord")), [tfrecord.make_example(
            {"id": i, "key": k.encode(), "source": s.tobytes(), "source_length": L, "text": b"abc"})])
        mel = g.normal(-40, 10, (T, 80)).astype("<f4")
        tfrecord.write_records(str(data / (k + ".target.tfrecord")), [tfrecord.make_example(
            {"id": i, "key": k.encode(), "mel": mel.tobytes(), "mel_width": 80, "target_length": T})])
    (lists / "train.csv").write_text("\n".join(keys[:4]) + "\n")
    (lists / "test.csv").write_text("\n".join(keys[4:]) + "\n")
    (lists / "validation.csv").write_text("\n".join(keys[2:4]) + "\n")
    import json
    d = json.load(open(os.path.join(ROOT, "examples", "ljspeech", "self-attention-tacotron.json")))
    d.pop("_comment", None)
    d.update(average_mel_level_db=[-40.0], stddev_mel_level_db=[10.0])       # the corpus statistics of preprocessing
    cfg = str(tmp_path / "hparams.json")
    json.dump(d, open(cfg, "w"))
    hp = ("batch_size=2,save_checkpoints_steps=2,alignment_save_steps=2,save_training_time_metrics=True,record_profile=True,"
          "profile_steps=3,logfile=%s" % (tmp_path / "log.txt"))
    common = ["--source-data-root", str(data), "--target-data-root", str(data), "--checkpoint-dir", str(ckpt),
              "--selected-list-dir", str(lists), "--hparam-json-file", cfg]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--max-steps", "4", "--hparams", hp] + common,
                       capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    log = open(tmp_path / "log.txt").read()
    assert os.path.exists(ckpt / "model-4.pt") and "step 4 loss" in log
    # TensorBoard scalars with the reference's names + the EVAL double pass after each checkpoint
    from satt_amd.utils.summary import read_events
    ev = [e for f in sorted(os.listdir(ckpt)) if f.startswith("events.out.tfevents") for e in read_events(str(ckpt / f))]
    steps = [e["step"] for e in ev if e["scalars"]]
    assert steps == [1, 2, 3, 4] and set(ev[-1]["scalars"]) == {"mel_loss", "done_loss", "loss", "learning_rate"}
    evd = str(ckpt / "eval")
    ee = [e for f in sorted(os.listdir(evd)) if f.startswith("events.out.tfevents")
          for e in read_events(os.path.join(evd, f)) if e["scalars"]]
    assert [e["step"] for e in ee] == [2, 4] and "eval step 4" in log
    assert set(ee[-1]["scalars"]) == {"mel_loss", "done_loss", "loss_with_teacher", "mel_loss_with_teacher",
                                      "done_loss_with_teacher"} and all(np.isfinite(v) for v in ee[-1]["scalars"].values())
    # training-time MetricsSaver dumps (alignment_save_steps, models/models.py:499-508), the EVAL-mode ones, and the profiler
    # hook's timeline of step 3 (record_profile / profile_steps, :510-513)
    dumps = sorted(f for f in os.listdir(ckpt) if f.startswith("train_result_step"))
    assert [f[:27] for f in dumps] == ["train_result_step000000001_", "train_result_step000000002_", "train_result_step000000004_"], dumps
    recs = [tfrecord.parse_prediction_result(p) for p in tfrecord.read_records(str(ckpt / dumps[-1]))]
    assert len(recs) == 2 and len(recs[0]["alignment"]) == 4                 # alignment1, alignment2, two encoder self-attention heads
    assert np.allclose(recs[0]["alignment"][0].sum(0), 1.0, atol=1e-4) and recs[0]["mel"].shape == recs[0]["ground_truth_mel"].shape
    assert any(f.startswith("alignment_step000000004_") and f.endswith(".png") for f in os.listdir(ckpt))
    assert any(f.startswith("eval_result_step000000004_") for f in os.listdir(evd))
    tl = json.load(open(ckpt / "timeline-3.json"))
    names = [e.get("name", "") for e in tl["traceEvents"]]
    assert any("attn_cluster_bwd_k" in n for n in names) and any("adam_k" in n for n in names)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "predict_mel.py"), "--output-dir", str(out), "--hparams",
                        "max_iters=12"] + common, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    for k in keys[4:]:
        mel = np.fromfile(out / (k + ".mfbsp"), dtype="<f4").reshape(-1, 80)
        al = np.load(out / (k + ".alignment.npz"))
        assert mel.shape[0] == 24 and np.isfinite(mel).all()                     # max_iters=12 steps x r=2 frames
        assert al["alignment"].shape[1] == 12 and np.allclose(al["alignment"].sum(0), 1.0, atol=1e-4)
        from satt_amd.utils.summary import read_png_size
        assert read_png_size(str(out / (k + ".png")))[1] == 12 * 4
        p = tfrecord.parse_prediction_result(next(tfrecord.read_records(str(out / (k + ".tfrecord")))))
        assert p["key"] == k and np.array_equal(p["mel"], mel) and len(p["alignment"]) == 2 and p["text"] == "abc"
        assert p["ground_truth_mel"].shape[1] == 80 and p["ground_truth_mel"].shape[0] > 0
    # forced-alignment mode (models/models.py:387-428): pass 1 = validation decode fed with the ground-truth mel, pass 2
    # = free feeding with both mechanisms returning pass 1's alignments: exactly Td = prepared_length / r steps
    out2 = tmp_path / "out2"; out2.mkdir()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "predict_mel.py"), "--output-dir", str(out2), "--hparams",
                        "max_iters=12,use_forced_alignment_mode=True"] + common, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    for k in keys[4:]:
        prepared = raw_len[k] + 4 + (raw_len[k] % 2)              # + 2r silence frames, tail-padded to a multiple of r = 2
        b = np.fromfile(out2 / (k + ".mfbsp"), dtype="<f4").reshape(-1, 80)
        al = np.load(out2 / (k + ".alignment.npz"))
        assert b.shape[0] == prepared and np.isfinite(b).all()
        assert al["alignment"].shape[1] == prepared // 2 and np.allclose(al["alignment"].sum(0), 1.0, atol=1e-4)
    # without the reference audio the mode cannot align to anything: refuse instead of silently free-running
    nc = [c for c in common]
    i = nc.index("--target-data-root"); del nc[i:i + 2]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "predict_mel.py"), "--output-dir", str(out2), "--hparams",
                        "use_forced_alignment_mode=True"] + nc, capture_output=True, text=True, timeout=200)
    assert r.returncode != 0 and "target-data-root" in (r.stderr + r.stdout)
    # resume: a second training run continues from model-4.pt (step counter, Adam moments) instead of starting over
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--max-steps", "6", "--hparams", hp] + common,
                       capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    log = open(tmp_path / "log.txt").read()
    assert "resumed from step 4" in log and os.path.exists(ckpt / "model-6.pt") and "step 6 loss" in log


def test_baseline_tacotron_train_then_predict(tmp_path):
    """examples/ljspeech/tacotron.json (ExtendedTacotronV1Model) through the same two command lines"""
    sys.path.insert(0, ROOT)
    import json
    import satt_amd  # noqa: F401
    from satt_amd.utils import tfrecord
    g = np.random.default_rng(1)
    data, lists, ckpt, out = tmp_path / "data", tmp_path / "lists", tmp_path / "ckpt", tmp_path / "out"
    for d in (data, lists, ckpt, out):
        d.mkdir()
    keys = ["LJ%03d" % i for i in range(5)]
    for i, k in enumerate(keys):
        L, T = int(g.integers(8, 16)), int(g.integers(20, 40))
        s = np.concatenate([[0], g.integers(1, 60, L - 2), [0]]).astype("<i8")
        tfrecord.write_records(str(data / (k + ".source.tfrecord")), [tfrecord.make_example(
            {"id": i, "key": k.encode(), "source": s.tobytes(), "source_length": L, "text": b"abc"})])
        mel = g.normal(-40, 10, (T, 80)).astype("<f4")
        tfrecord.write_records(str(data / (k + ".target.tfrecord")), [tfrecord.make_example(
            {"id": i, "key": k.encode(), "mel": mel.tobytes(), "mel_width": 80, "target_length": T})])
    (lists / "train.csv").write_text("\n".join(keys[:4]) + "\n")
    (lists / "test.csv").write_text(keys[4] + "\n")
    d = json.load(open(os.path.join(ROOT, "examples", "ljspeech", "tacotron.json")))
    d.pop("_comment", None)
    d.update(average_mel_level_db=[-40.0], stddev_mel_level_db=[10.0])
    cfg = str(tmp_path / "hparams.json")
    json.dump(d, open(cfg, "w"))
    hp = "batch_size=2,save_checkpoints_steps=3,logfile=%s" % (tmp_path / "log.txt")
    common = ["--source-data-root", str(data), "--target-data-root", str(data), "--checkpoint-dir", str(ckpt),
              "--selected-list-dir", str(lists), "--hparam-json-file", cfg]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--max-steps", "3", "--hparams", hp] + common,
                       capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.exists(ckpt / "model-3.pt") and "step 3 loss" in open(tmp_path / "log.txt").read()
    import torch
    n = torch.load(ckpt / "model-3.pt", map_location="cpu")["params"].numel()
    assert n < 6.0e6            # no self-attention blocks, no second mechanism: fewer parameters than the dual-source model
    r = subprocess.run([sys.executable, os.path.join(ROOT, "predict_mel.py"), "--output-dir", str(out), "--hparams",
                        "max_iters=10"] + common, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    mel = np.fromfile(out / (keys[4] + ".mfbsp"), dtype="<f4").reshape(-1, 80)
    al = np.load(out / (keys[4] + ".alignment.npz"))
    assert mel.shape[0] == 20 and np.isfinite(mel).all() and set(al.files) == {"alignment"}
    assert np.allclose(al["alignment"].sum(0), 1.0, atol=1e-4)
    p = tfrecord.parse_prediction_result(next(tfrecord.read_records(str(out / (keys[4] + ".tfrecord")))))
    assert len(p["alignment"]) == 1 and np.array_equal(p["mel"], mel)


def test_warm_start_from_a_tf_checkpoint(tmp_path):
    """hparams warm_start / ckpt_to_initialize_from / vars_to_warm_start (reference train.py:76-78, hparams.py:200-202)
    through tacotron_model_factory: a TF-format checkpoint (written by utils/tf_checkpoint.py under mapped names) initialises
    the selected variables of a fresh model; a checkpoint in model_dir takes precedence on the next start."""
    sys.path.insert(0, ROOT)
    import json
    import torch
    import satt_amd  # noqa: F401
    from satt_amd.hparams import hparams as default_hparams
    from satt_amd.models.models import tacotron_model_factory
    from satt_amd.models.warm_start import export_tf_checkpoint, template
    from satt_amd.params import ModelConfig
    hp = default_hparams.copy()
    d = json.load(open(os.path.join(ROOT, "examples", "ljspeech", "self-attention-tacotron.json"))); d.pop("_comment", None)
    hp.parse_json(json.dumps(d))
    src = tacotron_model_factory(hp, None, device="cuda", rng_seed=0)
    with torch.no_grad():
        src.engine.flat.add_(0.01 * torch.randn_like(src.engine.flat))
    vmap = {k.replace("?/", "model/"): v for k, v in template(ModelConfig.from_hparams(hp)).items() if k != "_comment"}
    mp = str(tmp_path / "map.json"); json.dump(vmap, open(mp, "w"))
    os.makedirs(tmp_path / "tf")
    export_tf_checkpoint(src.engine, str(tmp_path / "tf" / "model.ckpt-100"), vmap, global_step=100)
    (tmp_path / "tf" / "checkpoint").write_text('model_checkpoint_path: "model.ckpt-100"\n')
    hp2 = hp.copy()
    hp2.parse("warm_start=True,ckpt_to_initialize_from=%s,warm_start_var_map=%s" % (tmp_path / "tf", mp))
    hp2.vars_to_warm_start = ["model/dec\\."]
    run = str(tmp_path / "run")
    dst = tacotron_model_factory(hp2, run, device="cuda", rng_seed=0)
    P0, P1 = src.engine.P, dst.engine.P
    assert dst.warm_started and all(n.startswith("model/dec.") for n in dst.warm_started)
    assert torch.equal(P1["dec.lstm1.W"], P0["dec.lstm1.W"]) and torch.equal(P1["dec.sa.kvq.W"], P0["dec.sa.kvq.W"])
    assert not torch.equal(P1["enc.proj1.W"], P0["enc.proj1.W"]) and dst.global_step == 0
    # bf16 shadows follow the warm-started masters
    assert torch.equal(dst.engine.W("dec.lstm1.W").n, P0["dec.lstm1.W"].to(torch.bfloat16))
    dst.global_step = 3
    dst.save()
    again = tacotron_model_factory(hp2, run, device="cuda", rng_seed=0)      # model_dir has a checkpoint now: no warm start
    assert again.warm_started == [] and again.global_step == 3
    hp3 = hp.copy(); hp3.parse("warm_start=True,ckpt_to_initialize_from=%s" % (tmp_path / "tf"))
    with pytest.raises(ValueError, match="variable map"):
        tacotron_model_factory(hp3, str(tmp_path / "run3"), device="cuda")


def test_train_loop_skips_the_update_of_the_step_that_detects_a_handoff_timeout():
    """ADVICE r3 (models/models.py train): the iteration whose host check finds a hand-off timeout used to clear the error words
    and THEN run optimizer_step - the garbage gradients of that very step went into Adam while the warning said "skipped".
    Here the error word is set (by hand: what a bounded spin writes when it gives up) while batch 3 is being fetched, with
    log_step_count_steps=1 so the same iteration detects it: parameters and both moments must be exactly those after step 2,
    the loop must fall back to the chunked schedule, and training must continue from there."""
    sys.path.insert(0, ROOT)
    import json
    import torch
    import satt_amd  # noqa: F401
    from satt_amd.hparams import hparams as default_hparams
    from satt_amd.models.models import RunConfig, tacotron_model_factory
    from satt_amd.datasets.synthetic import synthetic_batch
    hp = default_hparams.copy()
    d = json.load(open(os.path.join(ROOT, "examples", "ljspeech", "self-attention-tacotron.json"))); d.pop("_comment", None)
    hp.parse_json(json.dumps(d))
    hp.parse("log_step_count_steps=1,save_checkpoints_steps=1000")
    model = tacotron_model_factory(hp, None, RunConfig.from_hparams(hp), device="cuda", rng_seed=0)
    eng = model.engine
    snap = {}

    def batches():
        for i in range(5):
            if i == 2:                         # steps 1 and 2 are done: remember the state, then make step 3's kernels "time out"
                torch.cuda.synchronize()
                snap.update(p=eng.flat.clone(), m=eng.m.clone(), v=eng.v.clone())
                eng._ws_last["attn"][-64:].view(torch.int32)[0] = 1
            if i == 3:                         # the iteration of batch 3 (index 2) is over: it must not have touched anything
                torch.cuda.synchronize()
                snap.update(p3=eng.flat.clone(), m3=eng.m.clone(), v3=eng.v.clone(), single=eng.single_launch_attention,
                            step=model.global_step)
            yield synthetic_batch(8, 40, 64, seed=10 + i, min_source_length=20, min_target_steps=16)
    model.train(batches)
    assert torch.equal(snap["p3"], snap["p"]) and torch.equal(snap["m3"], snap["m"]) and torch.equal(snap["v3"], snap["v"])
    assert snap["single"] is False and snap["step"] == 2          # fell back to the chunked schedule; the bad batch did not count
    assert model.global_step == 4 and not torch.equal(eng.flat, snap["p"])       # batches 4 and 5 trained on
    assert bool(torch.isfinite(eng.flat).all()) and bool(torch.isfinite(eng.m).all())


def test_train_raises_on_a_non_finite_gradient():
    """ADVICE r4 (csrc/elementwise.hip adam_prepare_k): the device-side guard skips the update of a step whose gradient is not
    finite, and nothing on the host ever looked - a diverged run logged `loss nan` forever.  The reference's Estimator aborts
    (NanLossDuringTrainingError); so does train() now, at the step's log point, with the parameters of the last good step."""
    sys.path.insert(0, ROOT)
    import json
    import torch
    import satt_amd  # noqa: F401
    from satt_amd.hparams import hparams as default_hparams
    from satt_amd.models.models import NanLossDuringTrainingError, RunConfig, tacotron_model_factory
    from satt_amd.datasets.synthetic import synthetic_batch
    hp = default_hparams.copy()
    d = json.load(open(os.path.join(ROOT, "examples", "ljspeech", "self-attention-tacotron.json"))); d.pop("_comment", None)
    hp.parse_json(json.dumps(d))
    hp.parse("log_step_count_steps=1,save_checkpoints_steps=1000")
    model = tacotron_model_factory(hp, None, RunConfig.from_hparams(hp), device="cuda", rng_seed=0)
    eng = model.engine
    snap = {}

    def batches():
        for i in range(4):
            b = synthetic_batch(8, 40, 64, seed=20 + i, min_source_length=20, min_target_steps=16)
            if i == 2:
                torch.cuda.synchronize()
                snap.update(p=eng.flat.clone(), m=eng.m.clone())
                b["mel"] = np.array(b["mel"], dtype=np.float32).copy()
                b["mel"][0, 3, 5] = np.nan          # one NaN target frame: loss and every gradient become NaN
            yield b
    with pytest.raises(NanLossDuringTrainingError):
        model.train(batches)
    assert model.global_step == 3                   # raised at the log point of the third step
    assert torch.equal(eng.flat, snap["p"]) and torch.equal(eng.m, snap["m"])       # ... whose update the device had skipped


def test_a_nan_step_between_two_log_steps_is_not_a_silently_dropped_update():
    """ADVICE r5 (models/models.py): the skip flag of adam_prepare_k is rewritten every step, and the host looked at log steps only -
    an isolated non-finite step in between was a dropped update nobody heard of.  The device now COUNTS its skips (sticky, behind
    the sum-of-squares partials); train() raises at the next log / checkpoint step and says how many updates were dropped."""
    sys.path.insert(0, ROOT)
    import json
    import torch
    import satt_amd  # noqa: F401
    from satt_amd.hparams import hparams as default_hparams
    from satt_amd.models.models import NanLossDuringTrainingError, RunConfig, tacotron_model_factory
    from satt_amd.datasets.synthetic import synthetic_batch
    hp = default_hparams.copy()
    d = json.load(open(os.path.join(ROOT, "examples", "ljspeech", "self-attention-tacotron.json"))); d.pop("_comment", None)
    hp.parse_json(json.dumps(d))
    hp.parse("log_step_count_steps=5,save_checkpoints_steps=1000")
    model = tacotron_model_factory(hp, None, RunConfig.from_hparams(hp), device="cuda", rng_seed=0)
    eng = model.engine

    def batches():
        for i in range(8):
            b = synthetic_batch(8, 40, 64, seed=20 + i, min_source_length=20, min_target_steps=16)
            if i == 1:          # step 2 of 5: not a log step
                b["mel"] = np.array(b["mel"], dtype=np.float32).copy()
                b["mel"][0, 3, 5] = np.nan
            yield b
    with pytest.raises(NanLossDuringTrainingError, match="1 update"):
        model.train(batches)
    assert model.global_step == 5                   # raised at the first log step behind the NaN step
    assert int(eng.opt_state[-2]) == 1 and int(eng.opt_state[-1]) == 1 and float(eng.opt_state[4]) == 0.0
    assert bool(torch.isfinite(eng.flat).all())     # the NaN gradient never reached the parameters


def test_poison_on_error_makes_the_skip_global():
    """ADVICE r3 (engine.py optimizer_step): the device-side skip was per rank.  satt_poison_on_error writes NaN into the first
    element of the last gradient bucket iff an error word is set, so the all-reduced gradient is non-finite on EVERY rank and
    satt_adam_step (which now also skips on a non-finite sum of squares) leaves all replicas untouched together."""
    sys.path.insert(0, ROOT)
    import torch
    import satt_amd  # noqa: F401
    from satt_amd import ops
    g = torch.randn(4096, device="cuda")
    err = torch.zeros(64, dtype=torch.uint8, device="cuda")
    ops.poison_on_error(g, [ops.cluster_err_word(err)])
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g).all())                           # healthy rank: untouched
    err.view(torch.int32)[0] = 1
    ops.poison_on_error(g, [None, ops.cluster_err_word(err)])
    torch.cuda.synchronize()
    assert bool(torch.isnan(g[0])) and bool(torch.isfinite(g[1:]).all())
    # what a HEALTHY peer sees after the sum: its own finite gradient + the poisoned one -> the update is skipped there too
    from satt_amd.engine import Engine
    from common import MEDIUM, make_params, small_batch
    cfg, P = make_params(MEDIUM, seed=4)
    eng = Engine(cfg, "cuda", params=P, rng_seed=3, lr0=2e-3, decay=False)
    b = eng.to_device_batch(small_batch(cfg, 8, 24, 40, seed=9))
    ctx = eng.train_step(b); eng.optimizer_step()
    p1, m1 = eng.flat.clone(), eng.m.clone()
    ctx = eng.train_step(b)
    eng.grad[:1] += g[:1]                                          # the all-reduce's contribution of the poisoned peer
    eng.optimizer_step()
    torch.cuda.synchronize()
    eng.check_clusters(ctx)                                        # this rank's own kernels were fine
    assert torch.equal(eng.flat, p1) and torch.equal(eng.m, m1)
    ctx = eng.train_step(b); eng.optimizer_step()                 # and a finite gradient updates again
    assert not torch.equal(eng.flat, p1)
