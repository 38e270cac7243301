"""TensorFlow-free observability / prediction outputs: TensorBoard event files, the
prediction record of the reference's predict driver (utils/tfrecord.py:135-152) and the alignment PNG."""
import os
import struct
import zlib

import numpy as np

import satt_amd  # noqa: F401
from satt_amd.utils import tfrecord
from satt_amd.utils.summary import (EventFileWriter, decode_event, encode_event, plot_alignments, read_events,
                                    read_png_size, write_png)


def test_event_encoding_known_bytes():
    """hand-assembled wire format of Event{wall_time=1.5, step=3, summary{value{tag='loss', simple_value=0.25}}}"""
    value = b"\x0a\x04loss" + b"\x15" + struct.pack("<f", 0.25)
    summary = b"\x0a" + bytes([len(value)]) + value
    expect = b"\x09" + struct.pack("<d", 1.5) + b"\x10\x03" + b"\x2a" + bytes([len(summary)]) + summary
    assert encode_event(1.5, 3, {"loss": 0.25}) == expect
    ev = decode_event(expect)
    assert ev["wall_time"] == 1.5 and ev["step"] == 3 and ev["scalars"] == {"loss": 0.25}


def test_event_file_round_trip_and_framing(tmp_path):
    w = EventFileWriter(str(tmp_path))
    for s in range(1, 4):
        w.add_scalars(s, {"loss": 1.0 / s, "mel_loss": 0.5 / s, "learning_rate": 1e-3}, wall_time=100.0 + s)
    w.close()
    evs = read_events(w.path)                       # read_records verifies both masked CRC-32C fields of every record
    assert evs[0]["file_version"] == "brain.Event:2" and evs[0]["step"] == 0
    assert [e["step"] for e in evs[1:]] == [1, 2, 3]
    assert abs(evs[2]["scalars"]["loss"] - 0.5) < 1e-7 and abs(evs[3]["scalars"]["mel_loss"] - 0.5 / 3) < 1e-7
    assert evs[1]["wall_time"] == 101.0
    assert w.path.split("/")[-1].startswith("events.out.tfevents.")


def test_prediction_record_round_trip(tmp_path):
    g = np.random.default_rng(0)
    mel = g.normal(size=(14, 80)).astype(np.float32)
    gt = g.normal(size=(16, 80)).astype(np.float32)
    al = [g.random((9, 7)).astype(np.float32), g.random((9, 7)).astype(np.float32)]
    src = np.arange(9, dtype=np.int64)
    f = str(tmp_path / "LJ001-0001.tfrecord")
    tfrecord.write_prediction_result(12, "LJ001-0001", al, mel, gt, "some text", src, None, f)
    recs = list(tfrecord.read_records(f))
    assert len(recs) == 1
    raw = tfrecord.parse_example(recs[0])
    assert set(raw) == {"id", "key", "mel", "mel_length", "mel_width", "ground_truth_mel", "ground_truth_mel_length",
                        "alignment", "text", "source", "source_length", "accent_type"}        # the reference's fields
    assert raw["accent_type"] == [] and len(raw["alignment"]) == 2
    p = tfrecord.parse_prediction_result(recs[0])
    assert p["id"] == 12 and p["key"] == "LJ001-0001" and p["text"] == "some text"
    assert np.array_equal(p["mel"], mel) and np.array_equal(p["ground_truth_mel"], gt) and np.array_equal(p["source"], src)
    assert all(np.array_equal(a, b) for a, b in zip(p["alignment"], al))


def test_png_writer(tmp_path):
    img = (np.arange(6 * 5).reshape(6, 5) * 8).astype(np.uint8)
    f = str(tmp_path / "g.png")
    write_png(f, img)
    assert read_png_size(f) == (6, 5)
    d = open(f, "rb").read()
    i = d.index(b"IDAT")
    n = struct.unpack(">I", d[i - 4:i])[0]
    raw = zlib.decompress(d[i + 4:i + 4 + n])
    rows = np.frombuffer(raw, np.uint8).reshape(6, 6)
    assert np.all(rows[:, 0] == 0) and np.array_equal(rows[:, 1:], img)         # filter byte 0 + the pixels
    a = np.random.default_rng(1).random((9, 7)).astype(np.float32)
    f2 = str(tmp_path / "a.png")
    plot_alignments(f2, [a, a], scale=3, gap=4)
    assert read_png_size(f2) == (2 * 9 * 3 + 4, 7 * 3)


def test_metrics_saver_dumps_and_prunes(tmp_path):
    """alignment_save_steps / save_training_time_metrics / keep_eval_results_max_epoch (reference models/models.py:499-508):
    result records carry the reference's prediction feature names, one alignment plot per utterance, old eval results go"""
    from satt_amd.utils.metrics_saver import MetricsSaver
    g = np.random.default_rng(0)
    B, Td, Ti, r, nm = 2, 6, 5, 2, 4
    al = [g.random((B, Td, Ti)).astype(np.float32), g.random((B, Td, Ti)).astype(np.float32)]
    mel, gt = g.normal(size=(B, Td * r, nm)).astype(np.float32), g.normal(size=(B, Td * r, nm)).astype(np.float32)
    src = g.integers(1, 9, (B, Ti)); sl = np.array([5, 3]); tl = np.array([12, 8])
    tr = MetricsSaver(str(tmp_path / "t"), 10, "train", save_training_time_metrics=False)
    assert not tr.due(10) and not tr.due(1)                                # hparams default: no training-time dumps
    tr = MetricsSaver(str(tmp_path / "t"), 10, "train", save_training_time_metrics=True)
    assert tr.due(10) and tr.due(1) and not tr.due(11)
    files = tr.save(10, [7, 9], ["k7", "k9"], ["a", "b"], src, sl, al, mel, gt, tl, r=r)
    assert os.path.basename(files[0]) == "train_result_step000000010_7,9.tfrecord"
    recs = [tfrecord.parse_prediction_result(p) for p in tfrecord.read_records(files[0])]
    assert [x["id"] for x in recs] == [7, 9] and recs[1]["key"] == "k9" and recs[1]["text"] == "b"
    assert recs[1]["mel"].shape == (8, nm) and recs[1]["ground_truth_mel"].shape == (8, nm) and len(recs[1]["source"]) == 3
    assert len(recs[1]["alignment"]) == 2 and recs[1]["alignment"][0].shape == (3, 4)       # [T_memory, T_query] of the valid part
    assert np.allclose(recs[1]["alignment"][1], al[1][1, :4, :3].T)
    assert all(os.path.exists(f) for f in files[1:]) and len(files) == 3
    ev = MetricsSaver(str(tmp_path / "e"), 10, "eval", keep_eval_results_max_epoch=2)
    for step in (10, 20, 30):
        assert ev.due(step)
        ev.save(step, [7, 9], ["k7", "k9"], ["a", "b"], src, sl, al, mel, gt, tl, r=r)
    left = sorted(os.listdir(tmp_path / "e"))
    assert not any("000000010" in f for f in left) and sum("000000030" in f for f in left) == 3
