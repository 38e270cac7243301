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
    W = target["mel"].shape[1]
    # the hparams defaults are [0.0] / [0.0] (reference hparams.py:20-21): the real tables come from the preprocessing run's
    # hparams.json.  (mel - 0) / 0 would silently train on inf / NaN targets, so a configuration without them is refused.
    if avg.size not in (1, W) or std.size not in (1, W):
        raise ValueError("average_mel_level_db / stddev_mel_level_db must have 1 or %d entries (got %d / %d)"
                         % (W, avg.size, std.size))
    if not np.all(std > 0):
        raise ValueError("stddev_mel_level_db contains zeros: pass the hparams.json written by the preprocessing run "
                         "(--hparam-json-file); the example configurations carry the model-selection keys only")
    mel = (target["mel"] - avg) / std
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


class BatchedDataset:
    """what `group_by_batch` returns (reference datasets/ljspeech/dataset.py:289-322): iterable of padded batch dicts with
    the fluent tail of the reference - `.prefetch(n)`, `.merge_target_to_source()`, `.dataset` - and usable directly as
    an iterator."""

    def __init__(self, make_iter, hparams):
        self._make, self._hparams, self._it = make_iter, hparams, None

    @property
    def hparams(self):
        return self._hparams

    @property
    def dataset(self):
        return self

    def __iter__(self):
        return self._make()

    def __next__(self):
        if self._it is None:
            self._it = self._make()
        return next(self._it)

    def prefetch(self, buffer_size):
        """batches are read, prepared and padded by a background thread, `buffer_size` of them ahead (:306-307): the
        record files of a batch are ~64 small reads, which would otherwise sit between two 10 ms GPU steps"""
        import queue
        import threading
        make = self._make
        n = max(1, int(buffer_size))

        def gen():
            q = queue.Queue(maxsize=n)
            END, stop = object(), threading.Event()

            def work():
                try:
                    for b in make():
                        while not stop.is_set():
                            try:
                                q.put(b, timeout=0.1)
                                break
                            except queue.Full:
                                continue
                        if stop.is_set():
                            return
                    q.put(END)
                except BaseException as e:          # surfaced in the consumer, never swallowed
                    q.put(e)
            th = threading.Thread(target=work, daemon=True)
            th.start()
            try:
                while True:
                    b = q.get()
                    if b is END:
                        return
                    if isinstance(b, BaseException):
                        raise b
                    yield b
            finally:
                stop.set()
        return BatchedDataset(gen, self._hparams)

    def merge_target_to_source(self):
        """prediction-time form (:309-322): the source side also carries mel / mel_width / target_length.  Batches here
        are flat dicts that already hold both sides, so this only guarantees the target fields are present."""
        make = self._make

        def gen():
            for b in make():
                if "mel" not in b or "target_length" not in b:
                    raise ValueError("merge_target_to_source: batch has no target fields")
                b = dict(b)
                b.setdefault("mel_width", b["mel"].shape[-1])
                yield b
        return BatchedDataset(gen, self._hparams)


class Dataset:
    """`dataset_factory(...).prepare_and_zip().filter_by_max_output_length().shuffle(n).group_by_batch(B)` of the
    reference (datasets/dataset_factory.py:12-35, datasets/ljspeech/dataset.py:112-115,194-216,235-286) as a plain
    Python pipeline over padded batch dicts.  source_files / target_files: parallel lists of record files; one record
    per file (the reference's `<key>.source.tfrecord` layout) unless built by `create_from_tfrecord_files`."""

    def __init__(self, source_files, target_files, hparams, cycle_length=None):
        if len(source_files) != len(target_files):
            raise ValueError("source and target file lists differ in length")
        self.files = list(zip(source_files, target_files))
        self.hparams = hparams
        self.cycle_length = cycle_length       # not None: every record of every file, interleaved (see below)
        self._filter = False
        self._shuffle = None
        self._repeat = False

    @staticmethod
    def create_from_tfrecord_files(source_files, target_files, hparams, cycle_length=4, buffer_output_elements=None,
                                   prefetch_input_elements=None):
        """reference datasets/ljspeech/dataset.py:94-110: files may hold MANY records; they are read `cycle_length`
        files at a time, one record from each in turn (tf.contrib.data.parallel_interleave, sloppy=False - a
        deterministic order).  The two buffer arguments tune tf.data's readers and have no effect on results."""
        return DatasetSource(source_files, target_files, hparams, cycle_length=max(1, int(cycle_length)))

    def prepare_and_zip(self):
        return self

    def filter_by_max_output_length(self):
        """drop utterances whose PREPARED target_length (raw + 2r, tail-padded to a multiple of r) exceeds
        max_iters * outputs_per_step (:197-202 runs after prepare_and_zip)"""
        self._filter = True
        return self

    def shuffle(self, buffer_size, seed=0):
        self._shuffle = (buffer_size, seed)
        return self

    def repeat(self):
        self._repeat = True
        return self

    def _pairs(self, order):
        hp = self.hparams
        if self.cycle_length is None:
            for i in order:
                s, m, _ = read_pair(*self.files[i], hp)
                yield s, m
            return
        # interleave: cycle_length slots, one (source, target) record pair from each slot per round; a slot whose file is
        # exhausted opens the next file on the spot (tf.data interleave order, sloppy=False)
        pending = [self.files[i] for i in order]

        def open_next():
            if not pending:
                return None
            sf, tf_ = pending.pop(0)
            return tfrecord.read_records(sf), tfrecord.read_records(tf_)
        slots = [open_next() for _ in range(self.cycle_length)]
        while any(sl is not None for sl in slots):
            for i in range(len(slots)):
                while slots[i] is not None:
                    rs, rt = slots[i]
                    ps, pt = next(rs, None), next(rt, None)
                    if (ps is None) != (pt is None):
                        raise ValueError("source and target files hold different numbers of records")
                    if ps is None:
                        slots[i] = open_next()
                        continue
                    yield decode_source_record(ps), prepare_target(decode_target_record(pt), hp)
                    break

    def _stream(self):
        hp = self.hparams
        epoch = 0
        while True:
            order = list(range(len(self.files)))
            if self._shuffle is not None:
                np.random.default_rng(self._shuffle[1] + epoch).shuffle(order)
            kept = 0
            for s, m in self._pairs(order):
                if self._filter and m.target_length > hp.max_iters * hp.outputs_per_step:
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

        def gen():
            buf = []
            for pair in self._stream():
                buf.append(pair)
                if len(buf) == bs:
                    yield pad_batch(buf, self.hparams)
                    buf = []
            if buf:
                yield pad_batch(buf, self.hparams)
        return BatchedDataset(gen, self.hparams)


class DatasetSource(Dataset):
    """the reference's class name (datasets/ljspeech/dataset.py:78, datasets/vctk/dataset.py: same reader; VCTK records
    add speaker_id / age / gender, which decode_source_record picks up when present)"""


DATASETS = ("ljspeech.dataset.DatasetSource", "vctk.dataset.DatasetSource")


def dataset_factory(source_files, target_files, hparams):
    """reference datasets/dataset_factory.py:12-18: `hparams.dataset` selects the class; both record layouts are handled
    by the same reader here."""
    if hparams.dataset not in DATASETS:
        raise ValueError("Unkown dataset")           # the reference's spelling (datasets/dataset_factory.py:18)
    return DatasetSource(source_files, target_files, hparams)


def create_from_tfrecord_files(source_files, target_files, hparams, cycle_length=4, buffer_output_elements=None,
                               prefetch_input_elements=None):
    """reference datasets/dataset_factory.py:21-35"""
    if hparams.dataset not in DATASETS:
        raise ValueError("Unkown dataset")
    return DatasetSource.create_from_tfrecord_files(source_files, target_files, hparams, cycle_length=cycle_length,
                                                    buffer_output_elements=buffer_output_elements,
                                                    prefetch_input_elements=prefetch_input_elements)
