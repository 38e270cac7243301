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


def test_length_filter_compares_the_prepared_length(tmp_path):
    """reference datasets/ljspeech/dataset.py:197-202 filters AFTER prepare_and_zip: raw + 2r (+ tail padding) frames"""
    h = hp(outputs_per_step=2, max_iters=10, average_mel_level_db=[0.0], stddev_mel_level_db=[1.0], batch_size=8)
    # max_iters * r = 20 prepared frames: raw 16 -> 20 (kept), raw 17 -> 22 (dropped), raw 15 -> 20 (19 padded; kept)
    src, tgt = _write_corpus(tmp_path, [(5, 16), (5, 17), (5, 15)])
    b = next(ljspeech.dataset_factory(src, tgt, h).prepare_and_zip().filter_by_max_output_length().group_by_batch())
    assert b["key"] == ["utt0", "utt2"] and b["target_length"].tolist() == [20, 20]


def test_fluent_tail_prefetch_and_merge_target_to_source(tmp_path):
    from satt_amd.datasets import dataset_factory as factory_module
    h = hp(outputs_per_step=2, batch_size=2, average_mel_level_db=[0.0], stddev_mel_level_db=[1.0])
    src, tgt = _write_corpus(tmp_path, [(5, 9), (8, 14), (3, 12), (6, 11), (4, 10)])
    plain = list(factory_module.dataset_factory(src, tgt, h).prepare_and_zip().group_by_batch())
    batched = factory_module.dataset_factory(src, tgt, h).prepare_and_zip().group_by_batch().prefetch(2)
    assert batched.dataset is batched and batched.hparams is h
    pre = list(batched.dataset)
    assert [b["key"] for b in pre] == [b["key"] for b in plain] == [["utt0", "utt1"], ["utt2", "utt3"], ["utt4"]]
    assert all(np.array_equal(a["mel"], b["mel"]) for a, b in zip(plain, pre))
    merged = next(iter(factory_module.dataset_factory(src, tgt, h).prepare_and_zip().group_by_batch(batch_size=1)
                       .merge_target_to_source()))
    assert merged["mel_width"] == 80 and merged["mel"].shape[0] == 1 and "target_length" in merged
    # a reader error in the background thread surfaces in the consumer
    os.remove(src[2])
    with pytest.raises(Exception):
        list(factory_module.dataset_factory(src, tgt, h).prepare_and_zip().group_by_batch().prefetch(1))
    h.dataset = "nope"
    with pytest.raises(ValueError, match="Unkown dataset"):
        factory_module.create_from_tfrecord_files(src, tgt, h)


def test_create_from_tfrecord_files_interleaves_multi_record_files(tmp_path):
    """reference datasets/ljspeech/dataset.py:94-110: parallel_interleave(cycle_length, sloppy=False) - one record from
    each of `cycle_length` open files in turn"""
    from satt_amd.datasets.dataset_factory import create_from_tfrecord_files
    h = hp(outputs_per_step=2, batch_size=16, average_mel_level_db=[0.0], stddev_mel_level_db=[1.0])
    g = np.random.default_rng(1)
    src, tgt, n = [], [], 0
    for f, count in enumerate([3, 1, 2]):
        rs, rt = [], []
        for j in range(count):
            key = ("f%dr%d" % (f, j)).encode()
            rs.append(tfrecord.make_example({"id": n, "key": key, "source": g.integers(1, 9, 4).astype("<i8").tobytes(),
                                             "source_length": 4, "text": b"t"}))
            mel = g.normal(0, 1, (6, 80)).astype("<f4")
            rt.append(tfrecord.make_example({"id": n, "key": key, "mel": mel.tobytes(), "mel_width": 80, "target_length": 6}))
            n += 1
        ps, pt = str(tmp_path / ("s%d.tfrecord" % f)), str(tmp_path / ("t%d.tfrecord" % f))
        tfrecord.write_records(ps, rs); tfrecord.write_records(pt, rt)
        src.append(ps); tgt.append(pt)
    b = next(create_from_tfrecord_files(src, tgt, h, cycle_length=2).prepare_and_zip().group_by_batch())
    assert b["key"] == ["f0r0", "f1r0", "f0r1", "f2r0", "f0r2", "f2r1"]
