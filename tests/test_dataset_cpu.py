"""CPU tests of the input contract: TFRecord / tf.train.Example codec without TensorFlow, target
preparation and batch padding (reference datasets/ljspeech/dataset.py:127-167,235-286, utils/tfrecord.py:43-104)."""
import copy
import os
import struct

import numpy as np
import pytest

import satt_amd  # noqa: F401
from satt_amd.datasets import ljspeech
from satt_amd.hparams import hparams as default_hparams
from satt_amd.utils import tfrecord


def hp(**kw):
    h = copy.deepcopy(default_hparams)
    h.parse("dataset=ljspeech.dataset.DatasetSource")
    for k, v in kw.items():
        setattr(h, k, v)
    return h


def test_crc32c_known_answers_and_mask():
    assert tfrecord.crc32c(b"123456789") == 0xE3069283          # the standard CRC-32C check value
    assert tfrecord.crc32c(b"") == 0
    assert tfrecord.crc32c(bytes(32)) == 0x8A9136AA              # RFC 3720 B.4: 32 bytes of zeros
    assert tfrecord.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43     # RFC 3720 B.4: 32 bytes of ones
    c = tfrecord.crc32c(b"abc")
    assert tfrecord.masked_crc(b"abc") == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_example_codec_round_trip_and_record_framing(tmp_path):
    mel = np.arange(12, dtype=np.float32).reshape(3, 4)
    ex = tfrecord.make_example({"id": 7, "key": b"LJ001-0001", "mel": mel.tobytes(), "mel_width": 4,
                                "target_length": 3, "ints": [-5, 2 ** 40, 0], "floats": np.array([1.5, -2.25], np.float32)})
    d = tfrecord.parse_example(ex)
    assert d["id"].tolist() == [7] and d["key"] == [b"LJ001-0001"] and d["ints"].tolist() == [-5, 2 ** 40, 0]
    assert np.array_equal(d["floats"], np.array([1.5, -2.25], np.float32))
    assert np.array_equal(np.frombuffer(d["mel"][0], "<f4").reshape(3, 4), mel)
    p = str(tmp_path / "a.tfrecord")
    tfrecord.write_records(p, [ex, b"second"])
    assert [bytes(r) for r in tfrecord.read_records(p)] == [ex, b"second"]
    raw = bytearray(open(p, "rb").read())
    assert struct.unpack("<Q", raw[:8])[0] == len(ex)
    raw[20] ^= 0x01                                             # flip one payload bit -> payload CRC must fail
    open(p, "wb").write(raw)
    with pytest.raises(tfrecord.TFRecordError):
        list(tfrecord.read_records(p))
    assert len(list(tfrecord.read_records(p, verify=False))) == 2


@pytest.mark.parametrize("T,want", [(5, 10), (6, 10), (7, 12)])
def test_prepare_target_contract(T, want):
    h = hp(outputs_per_step=2, average_mel_level_db=[1.0], stddev_mel_level_db=[2.0], silence_mel_level_db=-3.0)
    mel = np.arange(T * 3, dtype=np.float32).reshape(T, 3)
    m = ljspeech.prepare_target(dict(id=1, key="k", mel=mel, mel_width=3, target_length=T), h)
    assert m.target_length == want and m.mel.shape == (want, 3)
    assert np.all(m.mel[:2] == -3.0) and np.all(m.mel[2 + T:] == -3.0)           # r silence frames + tail padding
    assert np.allclose(m.mel[2:2 + T], (mel - 1.0) / 2.0)
    assert m.done.tolist() == [0.0] * (want // 2 - 1) + [1.0]
    assert m.spec_loss_mask.shape == (want,) and m.binary_loss_mask.shape == (want // 2,)
    assert m.spec_loss_mask.min() == 1.0 and m.binary_loss_mask.min() == 1.0


def _write_corpus(tmp_path, lengths, speakers=None):
    g = np.random.default_rng(0)
    src, tgt = [], []
    for i, (L, T) in enumerate(lengths):
        s = g.integers(1, 60, L).astype("<i8")
        f = {"id": i, "key": ("utt%d" % i).encode(), "source": s.tobytes(), "source_length": L, "text": b"hello"}
        if speakers:
            f.update(speaker_id=speakers[i], age=30, gender=1)
        ps, pt = str(tmp_path / ("utt%d.source.tfrecord" % i)), str(tmp_path / ("utt%d.target.tfrecord" % i))
        tfrecord.write_records(ps, [tfrecord.make_example(f)])
        mel = g.normal(0, 1, (T, 80)).astype("<f4")
        tfrecord.write_records(pt, [tfrecord.make_example({"id": i, "key": ("utt%d" % i).encode(), "mel": mel.tobytes(),
                                                           "mel_width": 80, "target_length": T})])
        src.append(ps); tgt.append(pt)
    return src, tgt


def test_dataset_batches_follow_the_padding_contract(tmp_path):
    h = hp(outputs_per_step=2, max_iters=20, silence_mel_level_db=-3.0, average_mel_level_db=[0.0],
           stddev_mel_level_db=[1.0], batch_size=2)
    src, tgt = _write_corpus(tmp_path, [(5, 9), (8, 14), (3, 50), (6, 11)])           # utt2 exceeds max_iters * r = 40
    ds = ljspeech.dataset_factory(src, tgt, h).prepare_and_zip().filter_by_max_output_length()
    batches = list(ds.group_by_batch())
    assert [len(b["key"]) for b in batches] == [2, 1] and batches[0]["key"] == ["utt0", "utt1"]
    b = batches[0]
    assert b["source"].shape == (2, 8) and b["source_length"].tolist() == [5, 8] and np.all(b["source"][0, 5:] == 0)
    assert b["target_length"].tolist() == [14, 18] and b["mel"].shape == (2, 18, 80)
    assert np.all(b["mel"][0, 14:] == -3.0) and np.all(b["done"][0, 6:] == 1.0) and b["done"][0, :6].sum() == 0.0
    assert b["spec_loss_mask"][0].tolist() == [1.0] * 14 + [0.0] * 4
    assert b["binary_loss_mask"][0].tolist() == [1.0] * 7 + [0.0] * 2
    assert "speaker_id" not in b


def test_vctk_records_carry_the_speaker_id(tmp_path):
    h = hp(outputs_per_step=2, batch_size=2, average_mel_level_db=[0.0], stddev_mel_level_db=[1.0])
    h.parse("dataset=vctk.dataset.DatasetSource")
    src, tgt = _write_corpus(tmp_path, [(5, 9), (4, 8)], speakers=[225, 376])
    b = next(ljspeech.dataset_factory(src, tgt, h).group_by_batch())
    assert b["speaker_id"].tolist() == [225, 376]
    h.dataset = "nope"
    with pytest.raises(ValueError):
        ljspeech.dataset_factory(src, tgt, h)
