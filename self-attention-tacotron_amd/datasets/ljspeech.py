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

Throughput (the 8.3 ms train step consumes ~3 850 utterances/s): record framing, both CRC-32C checks,
the tf.train.Example index and the target normalisation run in C (include/satt_io.h, GIL-free through ctypes); files are
read by a pool of `cycle_length` threads in DETERMINISTIC order (tf.contrib.data.parallel_interleave with sloppy=False,
reference datasets/ljspeech/dataset.py:100-109, train.py:34-36,46-49); the normalised mel is written straight into the
batch tensor (no per-utterance intermediate), which `.prefetch(n, pin_memory=True)` places in a ring of page-locked buffers.
"""
import os
from collections import deque, namedtuple

import numpy as np

from .. import _io
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


class RawTarget(namedtuple("RawTarget", ["id", "key", "raw_mel", "mel_width", "target_length", "raw_length", "lease"])):
    """a target record whose preparation is deferred to pad_batch (which writes it straight into the batch tensor):
    raw_mel = the record's float32 [raw_length, mel_width] array as stored, target_length = the PREPARED length
    (raw + 2r, rounded up to the next multiple of r).  `prepare()` gives the MelData the reference's map would.
    lease: None, or (owner, token) - raw_mel is a view of a pooled read buffer that `release()` hands back through
    owner.give_back(token) (the consumer calls it once the mel has been copied out; the view must not be touched afterwards)."""

    def release(self):
        if self.lease is not None:
            self.lease[0].give_back(self.lease[1])

    def prepare(self, hparams):
        return prepare_target(dict(id=self.id, key=self.key, mel=self.raw_mel, mel_width=self.mel_width,
                                   target_length=self.raw_length), hparams)


class _ArenaPool(deque):
    """free list of the Python reader's file-image buffers (deque.append / pop are atomic)"""
    give_back = deque.append


def _first(payload, idx, name):
    return _io.example_first_bytes(payload, idx[name])


def _int(payload, idx, name, default=0):
    e = idx.get(name)
    if e is None or e[2] == 0 or e[0] != 3:
        return default
    return int(e[5])                      # first value of the int64 list, decoded by satt_example_index


def decode_source_view(payload):
    """decode_source_record over the C index of the payload (bytes or memoryview)"""
    idx = _io.example_index(payload)
    src = np.frombuffer(_first(payload, idx, "source"), dtype="<i8").astype(np.int64)
    return SourceData(id=_int(payload, idx, "id"), key=bytes(_first(payload, idx, "key")).decode("utf-8"), source=src,
                      source_length=_int(payload, idx, "source_length"),
                      text=bytes(_first(payload, idx, "text")).decode("utf-8") if "text" in idx and idx["text"][2] else "",
                      speaker_id=_int(payload, idx, "speaker_id", -1), age=_int(payload, idx, "age", -1),
                      gender=_int(payload, idx, "gender", -1))


def decode_target_view(payload, r):
    """target record -> RawTarget: the mel stays a zero-copy float32 view of the record bytes"""
    idx = _io.example_index(payload)
    T, W = _int(payload, idx, "target_length"), _int(payload, idx, "mel_width")
    mel = np.frombuffer(_first(payload, idx, "mel"), dtype="<f4")
    if mel.size != T * W:
        raise ValueError("target record: mel holds %d floats, target_length x mel_width = %d x %d" % (mel.size, T, W))
    return RawTarget(_int(payload, idx, "id"), bytes(_first(payload, idx, "key")).decode("utf-8"), mel.reshape(T, W), W,
                     _io.prepared_length(T, r), T, None)


def _norm_tables(hparams, W):
    avg = np.ascontiguousarray(hparams.average_mel_level_db, dtype=np.float32).reshape(-1)
    std = np.ascontiguousarray(hparams.stddev_mel_level_db, dtype=np.float32).reshape(-1)
    # the hparams defaults are [0.0] / [0.0] (reference hparams.py:20-21): the real tables come from the preprocessing run's
    # hparams.json.  (mel - 0) / 0 would silently train on inf / NaN targets, so a configuration without them is refused.
    if avg.size not in (1, W) or std.size not in (1, W):
        raise ValueError("average_mel_level_db / stddev_mel_level_db must have 1 or %d entries (got %d / %d)"
                         % (W, avg.size, std.size))
    if not np.all(std > 0):
        raise ValueError("stddev_mel_level_db contains zeros: pass the hparams.json written by the preprocessing run "
                         "(--hparam-json-file); the example configurations carry the model-selection keys only")
    return avg, std


def prepare_target(target, hparams):
    """raw target record -> MelData (reference datasets/ljspeech/dataset.py:127-167)"""
    r = hparams.outputs_per_step
    sil = np.float32(hparams.silence_mel_level_db)
    W = target["mel"].shape[1]
    avg, std = _norm_tables(hparams, W)
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
    s = decode_source_view(tfrecord.read_record_views(source_file)[0])
    t = decode_target_view(tfrecord.read_record_views(target_file)[0], hparams.outputs_per_step)
    return s, t.prepare(hparams), t.raw_length


class PinnedRing:
    """`slots` reusable sets of page-locked batch buffers (hipHostMalloc of an 8 MB batch costs milliseconds: never per
    batch).  A slot is handed out again after `slots - 1` further batches, so a consumer may hold a batch while at most
    slots - 2 newer ones are drawn; `.prefetch(n, pin_memory=True)` sizes the ring n + 4.  Falls back to ordinary memory
    where no GPU runtime is present (CPU tests)."""

    def __init__(self, slots):
        self.slots, self.bufs, self.next = max(2, int(slots)), {}, 0
        self.fences = {}          # slot -> event behind the asynchronous upload of the batch it held last
        try:
            import torch
            self.torch = torch if torch.cuda.is_available() else None
        except ImportError:
            self.torch = None

    def take(self):
        """the next slot to fill.  If its previous batch was uploaded asynchronously (Engine.to_device_batch recorded a fence
        through `uploaded`), wait for that copy first: the host may run several steps ahead of the GPU, and the queued H2D copy
        must have read the buffers before the prefetch thread overwrites them."""
        k = self.next
        self.next = (k + 1) % self.slots
        ev = self.fences.pop(k, None)
        if ev is not None:
            ev.synchronize()
        return k

    def uploaded(self, slot):
        """called by the consumer right behind the asynchronous upload of the slot's arrays (on the uploading stream)"""
        if self.torch is not None:
            ev = self.torch.cuda.Event()
            ev.record()
            self.fences[slot] = ev

    def array(self, slot, name, shape, dtype):
        """a [shape] view of the slot's buffer `name`, grown (never shrunk) to the largest batch seen"""
        n = int(np.prod(shape))
        cur = self.bufs.get((slot, name))
        if cur is None or cur[1].size < n or cur[1].dtype != dtype:
            cap = max(n, 1)
            if self.torch is not None:
                t = self.torch.empty(cap, dtype=getattr(self.torch, np.dtype(dtype).name), pin_memory=True)
                cur = (t, t.numpy())
            else:
                cur = (None, np.empty(cap, dtype))
            self.bufs[(slot, name)] = cur
        return cur[1][:n].reshape(shape)


class PinnedBatch(dict):
    """a batch dict whose arrays live in a PinnedRing slot; `.pinned` = (ring, slot) - not a key: the batch looks like any other"""
    pinned = None


def pad_batch(pairs, hparams, ring=None):
    """list of (SourceData, MelData | RawTarget) -> the engine's batch dict (padding values of group_by_batch, :264-281).
    RawTarget elements are normalised and silence-padded straight into their row of the batch tensor (satt_prepare_mel);
    ring (PinnedRing): the arrays live in page-locked memory (torch.as_tensor(a).is_pinned(): asynchronous H2D copies)."""
    B = len(pairs)
    Ti = max(len(s.source) for s, _ in pairs)
    Tm = max(m.target_length for _, m in pairs)
    r = hparams.outputs_per_step
    W = pairs[0][1].raw_mel.shape[1] if isinstance(pairs[0][1], RawTarget) else pairs[0][1].mel.shape[1]
    sil = np.float32(hparams.silence_mel_level_db)
    if ring is not None:
        slot = ring.take()
        new = lambda name, shape, dtype: ring.array(slot, name, shape, dtype)
    else:
        new = lambda name, shape, dtype: np.empty(shape, dtype)
    source = new("source", (B, Ti), np.int64); source.fill(0)
    mel = new("mel", (B, Tm, W), np.float32)
    done = new("done", (B, Tm // r), np.float32); done.fill(1.0)
    smask = new("spec_loss_mask", (B, Tm), np.float32); smask.fill(0.0)
    bmask = new("binary_loss_mask", (B, Tm // r), np.float32); bmask.fill(0.0)
    tables = None
    for b, (s, m) in enumerate(pairs):
        source[b, :len(s.source)] = s.source
        L = m.target_length
        if isinstance(m, RawTarget):
            if tables is None:
                tables = _norm_tables(hparams, W)
            raw = m.raw_mel if (m.raw_mel.flags.c_contiguous and m.raw_mel.dtype == np.float32) else \
                np.ascontiguousarray(m.raw_mel, np.float32)
            got = _io.prepare_mel(raw, tables[0], tables[1], r, sil, mel[b])           # whole row incl. the batch padding
            assert got == L
            m.release()
            done[b, :L // r - 1] = 0.0
            smask[b, :L] = 1.0
            bmask[b, :L // r] = 1.0
        else:
            mel[b, :L] = m.mel
            mel[b, L:] = sil
            done[b, :len(m.done)] = m.done
            smask[b, :L] = m.spec_loss_mask
            bmask[b, :len(m.binary_loss_mask)] = m.binary_loss_mask
    batch = dict(source=source, source_length=np.array([s.source_length for s, _ in pairs], np.int64), mel=mel,
                 target_length=np.array([m.target_length for _, m in pairs], np.int64), done=done,
                 spec_loss_mask=smask, binary_loss_mask=bmask,
                 id=np.array([s.id for s, _ in pairs], np.int64), key=[s.key for s, _ in pairs],
                 text=[s.text for s, _ in pairs])
    if pairs[0][0].speaker_id >= 0:
        batch["speaker_id"] = np.array([s.speaker_id for s, _ in pairs], np.int64)
    if ring is not None:
        batch = PinnedBatch(batch)
        batch.pinned = (ring, slot)          # lease of the page-locked slot: Engine.to_device_batch fences its asynchronous upload
    return batch


def get_parallelism(factor, min_value, max_value):
    """reference train.py:101-102"""
    return min(max(int((os.cpu_count() or 1) * factor), min_value), max_value)


def ordered_parallel_map(fn, items, workers, window, pool=None):
    """fn over items on `workers` threads, results in INPUT order, at most `window` items in flight (the deterministic
    order of parallel_interleave(sloppy=False)).  workers <= 1: a plain sequential map."""
    if workers <= 1:
        for x in items:
            yield fn(x)
        return
    from concurrent.futures import ThreadPoolExecutor
    own = pool is None
    ex = pool or ThreadPoolExecutor(max_workers=workers, thread_name_prefix="satt-reader")
    pending = deque()
    try:
        for x in items:
            pending.append(ex.submit(fn, x))
            if len(pending) >= max(workers, window):
                yield pending.popleft().result()
        while pending:
            yield pending.popleft().result()
    finally:
        for f in pending:
            f.cancel()
        if own:
            ex.shutdown(wait=False)


class BatchedDataset:
    """what `group_by_batch` returns (reference datasets/ljspeech/dataset.py:289-322): iterable of padded batch dicts with
    the fluent tail of the reference - `.prefetch(n)`, `.merge_target_to_source()`, `.dataset` - and usable directly as
    an iterator."""

    def __init__(self, make_iter, hparams, make_pinned=None):
        self._make, self._hparams, self._it = make_iter, hparams, None
        self._make_pinned = make_pinned         # make_pinned(ring) -> iterator whose batches live in the ring's buffers

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

    def prefetch(self, buffer_size, pin_memory=False):
        """batches are read, prepared and padded by a background thread, `buffer_size` of them ahead (:306-307): the
        record files of a batch are ~64 small reads, which would otherwise sit between two 8 ms GPU steps.
        pin_memory=True: the batch tensors are filled in a ring of buffer_size + 4 page-locked buffer sets (PinnedRing): a
        batch stays valid until buffer_size + 2 newer ones have been drawn - enough for a train loop that uploads each batch
        before it asks for the next, not for a consumer that collects batches."""
        import queue
        import threading
        n = max(1, int(buffer_size))
        make = self._make
        if pin_memory and self._make_pinned is not None:
            ring = PinnedRing(n + 4)
            make = lambda: self._make_pinned(ring)

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

    def __init__(self, source_files, target_files, hparams, cycle_length=None, buffer_output_elements=None,
                 prefetch_input_elements=None, num_workers=None):
        if len(source_files) != len(target_files):
            raise ValueError("source and target file lists differ in length")
        self.files = list(zip(source_files, target_files))
        self.hparams = hparams
        self.cycle_length = cycle_length       # not None: every record of every file, interleaved (see below)
        # reader threads: the interleave's cycle_length where one is given (parallel_interleave reads its cycle_length files
        # concurrently), otherwise the reference's get_parallelism(...) of the interleave hparams (train.py:34-36)
        self.num_workers = int(num_workers) if num_workers is not None else int(cycle_length) if cycle_length else \
            get_parallelism(getattr(hparams, "interleave_cycle_length_cpu_factor", 1.0),
                            getattr(hparams, "interleave_cycle_length_min", 4), getattr(hparams, "interleave_cycle_length_max", 16))
        # files in flight ahead of the consumer (prefetch_input_elements of parallel_interleave; records of a file are whole
        # in memory once it is read, which is what buffer_output_elements bounds there)
        self.window = max(self.num_workers, min(int(prefetch_input_elements or 2 * self.num_workers), 4 * self.num_workers))
        self._filter = False
        self._shuffle = None
        self._repeat = False
        self._cache = None
        self._arena_pool = _ArenaPool()
        self.native_reader = True              # satt_reader (POSIX threads) for one-record files; False: the Python thread pool
        self._lease_hint = int(getattr(hparams, "batch_size", 32))

    @staticmethod
    def create_from_tfrecord_files(source_files, target_files, hparams, cycle_length=4, buffer_output_elements=None,
                                   prefetch_input_elements=None):
        """reference datasets/ljspeech/dataset.py:94-110: files may hold MANY records; they are read `cycle_length`
        files at a time - concurrently, by that many threads - one record from each in turn
        (tf.contrib.data.parallel_interleave, sloppy=False: a deterministic order)."""
        return DatasetSource(source_files, target_files, hparams, cycle_length=max(1, int(cycle_length)),
                             buffer_output_elements=buffer_output_elements, prefetch_input_elements=prefetch_input_elements)

    def prepare_and_zip(self):
        return self

    def cache(self, filename=""):
        """reference train.py:51 (`zipped.cache(hparams.cache_file_name) if hparams.use_cache`): the decoded records of the
        first pass are kept - in memory here whatever `filename` says (LJSpeech's mels are ~3.3 GB) - and later epochs
        read no files."""
        self._cache = {}
        return self

    def filter_by_max_output_length(self):
        """drop utterances whose PREPARED target_length (raw + 2r, tail-padded to a multiple of r) exceeds
        max_iters * outputs_per_step (:197-202 runs after prepare_and_zip)"""
        self._filter = True
        return self

    def shuffle(self, buffer_size, seed=0):
        self._shuffle = (buffer_size, seed)
        return self

    def repeat(self, count=None):
        self._repeat = True
        return self

    def _load_file(self, i):
        """reader-thread body: every record pair of file pair i as (SourceData, RawTarget) - framing + CRC + Example index
        in C, the mel a zero-copy view of the file image"""
        if self._cache is not None and i in self._cache:
            return self._cache[i]
        sf, tf_ = self.files[i]
        r = self.hparams.outputs_per_step
        # read buffers come from a free list (a fresh 400 KB allocation per utterance is an mmap + munmap pair: with several
        # reader threads the TLB shootdowns of the unmaps cost more than the reads); cached records keep theirs
        pool = self._arena_pool if self._cache is None else None
        try:
            buf = pool.pop() if pool else None
        except IndexError:
            buf = None
        try:
            arena, u = _io.utterance_load(sf, tf_, r, arena=buf)
        except ValueError as e:
            raise tfrecord.TFRecordError(str(e)) from None
        if u.src_records == 1 and u.tgt_records == 1:      # the reference's layout: one record per file
            out = [self._utterance_pair(u, arena, (pool, arena) if pool is not None else None)]
        else:                                              # many records per file: the general reader
            out = self._load_general(i)
        if self._cache is not None:
            self._cache[i] = out
        return out

    def _utterance_pair(self, u, arena, lease):
        """(SourceData, RawTarget) from a decoded satt_utterance and its file images"""
        mv = memoryview(arena)
        T, W = int(u.target_length), int(u.mel_width)
        key = bytes(mv[u.key_off:u.key_off + u.key_len]).decode("utf-8")
        s_ = SourceData(id=int(u.id), key=key,
                        source=np.frombuffer(mv[u.source_off:u.source_off + 8 * u.source_count], dtype="<i8").astype(np.int64),
                        source_length=int(u.source_length),
                        text=bytes(mv[u.text_off:u.text_off + u.text_len]).decode("utf-8") if u.text_len else "",
                        speaker_id=int(u.speaker_id), age=int(u.age), gender=int(u.gender))
        mel = np.frombuffer(mv[u.mel_off:u.mel_off + 4 * u.mel_count], dtype="<f4").reshape(T, W)
        return s_, RawTarget(int(u.target_id), key, mel, W, int(u.prepared_length), T, lease)

    def _load_general(self, i):
        """every record pair of file pair i through the general (many records per file) reader"""
        sf, tf_ = self.files[i]
        rs, rt = tfrecord.read_record_views(sf), tfrecord.read_record_views(tf_)
        if len(rs) != len(rt):
            raise ValueError("source and target files hold different numbers of records")
        r = self.hparams.outputs_per_step
        return [(decode_source_view(ps), decode_target_view(pt, r)) for ps, pt in zip(rs, rt)]

    def _native_loaded(self, order):
        """_load_file over `order` on the native reader (include/satt_io.h satt_reader_*): POSIX threads, no GIL hand-offs,
        results in order.  Slots = look-ahead window + the utterances a consumer holds (two batches)."""
        rd = _io.Reader(self.num_workers, self.window + 2 * self._lease_hint + 2, self.hparams.outputs_per_step)
        it = iter(order)
        queued = deque()
        held = None
        try:
            while True:
                while len(queued) < self.window:
                    i = held if held is not None else next(it, None)
                    held = None
                    if i is None:
                        break
                    if self._cache is not None and i in self._cache:
                        queued.append((i, None))
                        continue
                    t = rd.submit(*self.files[i])
                    if t is None:                 # every slot busy or leased
                        held = i
                        break
                    queued.append((i, t))
                if not queued:
                    if held is None:
                        return
                    # all slots are leased to the consumer and nothing is in flight: read this one on the spot, unleased
                    recs = self._load_file(held)
                    held = None
                    yield recs
                    continue
                i, t = queued.popleft()
                if t is None:
                    yield self._cache[i]
                    continue
                t2, status, u, arena = rd.next()
                assert t2 == t
                if status != 0:
                    rd.give_back(t)
                    if status == -8:          # SATT_IO_E_IO: a file could not be opened or read (the errno stays in the reader thread)
                        bad = next((q for q in self.files[i] if not os.path.exists(q)), None)
                        if bad is not None:
                            raise FileNotFoundError(bad)
                        for q in self.files[i]:   # exists but unreadable: let the OS name the reason (EACCES, EMFILE, EIO ...)
                            with open(q, "rb") as f:
                                f.read(1)
                        raise OSError("%s / %s: cannot be opened or read" % self.files[i])
                    raise tfrecord.TFRecordError("%s / %s: %s" % (self.files[i] + (_io.ERRORS.get(status, "error %d" % status),)))
                if u.src_records == 1 and u.tgt_records == 1:
                    if self._cache is not None:      # cached records own their memory
                        s_, t_ = self._utterance_pair(u, arena, None)
                        recs = [(s_, t_._replace(raw_mel=t_.raw_mel.copy()))]
                        rd.give_back(t)
                        self._cache[i] = recs
                    else:
                        recs = [self._utterance_pair(u, arena, (rd, t))]
                else:
                    rd.give_back(t)
                    recs = self._load_general(i)
                    if self._cache is not None:
                        self._cache[i] = recs
                yield recs
        finally:
            del rd                                 # destroyed once the last leased RawTarget is gone too

    def _pairs(self, order):
        if self.native_reader and self.num_workers >= 1:
            loaded = self._native_loaded(order)
        else:
            loaded = ordered_parallel_map(self._load_file, order, self.num_workers, self.window)
        if self.cycle_length is None:
            for recs in loaded:
                if recs:                          # dataset_factory layout: one record per file (the first, as TFRecordDataset
                    yield recs[0]                 # zipped over single-record files yields)
            return
        # interleave: cycle_length slots, one (source, target) record pair from each slot per round; a slot whose file is
        # exhausted opens the next file on the spot (tf.data interleave order, sloppy=False)

        def open_next():
            recs = next(loaded, None)
            return None if recs is None else iter(recs)
        slots = [open_next() for _ in range(self.cycle_length)]
        while any(sl is not None for sl in slots):
            for i in range(len(slots)):
                while slots[i] is not None:
                    pair = next(slots[i], None)
                    if pair is None:
                        slots[i] = open_next()
                        continue
                    yield pair
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
                    if isinstance(m, RawTarget):
                        m.release()
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
        self._lease_hint = int(bs)

        def gen(ring=None):
            buf = []
            for pair in self._stream():
                buf.append(pair)
                if len(buf) == bs:
                    yield pad_batch(buf, self.hparams, ring)
                    buf = []
            if buf:
                yield pad_batch(buf, self.hparams, ring)
        return BatchedDataset(gen, self.hparams, make_pinned=gen)

    def __iter__(self):
        """the zipped elements as the reference's (SourceData, MelData) namedtuples (prepared targets)"""
        for s, m in self._stream():
            if isinstance(m, RawTarget):
                p = m.prepare(self.hparams)
                m.release()
                m = p
            yield s, m


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
