"""Input contract of the training path without tf.data: utterance records -> padded batches.

Mirrors the behaviour of the reference's dataset classes (datasets/ljspeech/dataset.py:52-72 record fields,
:127-167 `_prepare_target`, :194-202 length filter, :235-286 `group_by_batch`; datasets/vctk/dataset.py:36-38,70-72
for the extra speaker fields) on plain numpy:
  * target: mel normalised `(mel - average_mel_level_db) / stddev_mel_level_db`, `r` silence frames (value
    `silence_mel_level_db`) in front and behind, `target_length += 2r`, tail-padded with silence up to the NEXT
    multiple of r when the length is not one already, `done = [0 ... 0 1]` (length T/r), loss masks of ones;
  * batch: source padded with 0, mel with `silence_mel_level_db`, done with 1, loss masks with 0.
The reference's bucketing key is `min(target_length - approx_min_target_length, 0) // batch_bucket_width`, i.e.
<= 0 for every utterance: batches are effectively unbucketed windows, so batching here is
"next `batch_size` utterances of the (shuffled) stream".
"""
from collections import namedtuple

import numpy as np

from ..utils import tfrecord

SourceData = namedtuple("SourceData", ["id", "key", "source", "source_length", "text", "speaker_id", "age", "gender"])
MelData = namedtuple("MelData", ["id", "key", "mel", "mel_width", "target_length", "done", "spec_loss_mask",
                                 "binary_loss_mask"])


def _scalar(v, default=0):
    return int(v[0]) if len(v) else default


def decode_source_record(payload):
    """`<key>.source.tfrecord` payload: id, key, source (raw int64 bytes), source_length, text [, speaker_id, age, gender]"""
    f = tfrecord.parse_example(payload)
    src = np.frombuffer(f["source"][0], dtype="<i8").astype(np.int64)
    return SourceData(id=_scalar(f["id"]), key=f["key"][0].decode("utf-8"), source=src,
                      source_length=_scalar(f["source_length"]), text=f["text"][0].decode("utf-8") if "text" in f else "",
                      speaker_id=_scalar(f.get("speaker_id", []), -1), age=_scalar(f.get("age", []), -1),
                      gender=_scalar(f.get("gender", []), -1))


def decode_target_record(payload):
    """`<key>.target.tfrecord` payload: id, key, mel (raw float32 bytes [target_length, mel_width]), mel_width,
    target_length (the spec / spec_width fields of the full target record are ignored, as in the mel-only parser)"""
    f = tfrecord.parse_example(payload)
    T, W = _scalar(f["target_length"]), _scalar(f["mel_width"])
    mel = np.frombuffer(f["mel"][0], dtype="<f4").astype(np.float32).reshape(T, W)
    return dict(id=_scalar(f["id"]), key=f["key"][0].decode("utf-8"), mel=mel, mel_width=W, target_length=T)


def prepare_target(target, hparams):
    """raw target record -> MelData (reference datasets/ljspeech/dataset.py:127-167)"""
    r = hparams.outputs_per_step
    sil = np.float32(hparams.silence_mel_level_db)
    avg = np.asarray(hparams.average_mel_level_db, dtype=np.float32)
    std = np.asarray(hparams.stddev_mel_level_db, dtype=np.float32)
    mel = (target["mel"] - avg) / std
    W = mel.shape[1]
    pad = np.full((r, W), sil, dtype=np.float32)
    mel = np.concatenate([pad, mel, pad], axis=0)
    length = target["target_length"] + 2 * r
    if length % r != 0:
        padded = (length // r + 1) * r
        mel = np.concatenate([mel, np.full((padded - length, W), sil, dtype=np.float32)], axis=0)
        length = padded
    done = np.concatenate([np.zeros(length // r - 1, np.float32), np.ones(1, np.float32)])
    return MelData(target["id"], target["key"], mel.astype(np.float32), target["mel_width"], length, done,
                   np.ones(length, np.float32), np.ones(length // r, np.float32))


def read_pair(source_file, target_file, hparams):
    """one utterance: (SourceData, MelData) from its two record files"""
    s = decode_source_record(next(tfrecord.read_records(source_file)))
    t = decode_target_record(next(tfrecord.read_records(target_file)))
    return s, prepare_target(t, hparams), t["target_length"]


def pad_batch(pairs, hparams):
    """list of (SourceData, MelData) -> the engine's batch dict (padding values of group_by_batch, :264-281)"""
    B = len(pairs)
    Ti = max(len(s.source) for s, _ in pairs)
    Tm = max(m.target_length for _, m in pairs)
    r, W = hparams.outputs_per_step, pairs[0][1].mel.shape[1]
    source = np.zeros((B, Ti), np.int64)
    mel = np.full((B, Tm, W), np.float32(hparams.silence_mel_level_db), np.float32)
    done = np.ones((B, Tm // r), np.float32)
    smask = np.zeros((B, Tm), np.float32)
    bmask = np.zeros((B, Tm // r), np.float32)
    for b, (s, m) in enumerate(pairs):
        source[b, :len(s.source)] = s.source
        mel[b, :m.target_length] = m.mel
        done[b, :len(m.done)] = m.done
        smask[b, :m.target_length] = m.spec_loss_mask
        bmask[b, :len(m.binary_loss_mask)] = m.binary_loss_mask
    batch = dict(source=source, source_length=np.array([s.source_length for s, _ in pairs], np.int64), mel=mel,
                 target_length=np.array([m.target_length for _, m in pairs], np.int64), done=done,
                 spec_loss_mask=smask, binary_loss_mask=bmask,
                 id=np.array([s.id for s, _ in pairs], np.int64), key=[s.key for s, _ in pairs],
                 text=[s.text for s, _ in pairs])
    if pairs[0][0].speaker_id >= 0:
        batch["speaker_id"] = np.array([s.speaker_id for s, _ in pairs], np.int64)
    return batch


class Dataset:
    """`dataset_factory(...).prepare_and_zip().filter_by_max_output_length().shuffle(n).group_by_batch(B)` of the
    reference (datasets/dataset_factory.py:12-35, datasets/ljspeech/dataset.py:112-115,194-216,235-286) as a plain
    Python iterator over padded batch dicts.  source_files / target_files: parallel lists of record files."""

    def __init__(self, source_files, target_files, hparams):
        if len(source_files) != len(target_files):
            raise ValueError("source and target file lists differ in length")
        self.files = list(zip(source_files, target_files))
        self.hparams = hparams
        self._filter = False
        self._shuffle = None
        self._repeat = False

    def prepare_and_zip(self):
        return self

    def filter_by_max_output_length(self):
        """drop utterances whose RAW target_length exceeds max_iters * outputs_per_step (:197-202)"""
        self._filter = True
        return self

    def shuffle(self, buffer_size, seed=0):
        self._shuffle = (buffer_size, seed)
        return self

    def repeat(self):
        self._repeat = True
        return self

    def _stream(self):
        hp = self.hparams
        epoch = 0
        while True:
            order = list(range(len(self.files)))
            if self._shuffle is not None:
                np.random.default_rng(self._shuffle[1] + epoch).shuffle(order)
            kept = 0
            for i in order:
                s, m, raw_len = read_pair(*self.files[i], hp)
                if self._filter and raw_len > hp.max_iters * hp.outputs_per_step:
                    continue
                kept += 1
                yield s, m
            if not self._repeat:
                return
            if kept == 0:
                raise ValueError("every utterance was filtered out (max_iters * outputs_per_step = %d frames)"
                                 % (hp.max_iters * hp.outputs_per_step))
            epoch += 1

    def group_by_batch(self, batch_size=None):
        bs = batch_size if batch_size is not None else self.hparams.batch_size
        buf = []
        for pair in self._stream():
            buf.append(pair)
            if len(buf) == bs:
                yield pad_batch(buf, self.hparams)
                buf = []
        if buf:
            yield pad_batch(buf, self.hparams)


def dataset_factory(source_files, target_files, hparams):
    """reference datasets/dataset_factory.py:12-35: `hparams.dataset` selects the class; both the LJSpeech and the VCTK
    record layouts are handled by the same reader here (VCTK adds speaker_id / age / gender)."""
    if hparams.dataset not in ("ljspeech.dataset.DatasetSource", "vctk.dataset.DatasetSource"):
        raise ValueError("Unknown dataset: %s" % hparams.dataset)
    return Dataset(source_files, target_files, hparams)
