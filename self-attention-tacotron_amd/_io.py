"""ctypes binding of libsatt_io.so (include/satt_io.h): the host-side C of the input pipeline.  ctypes drops the GIL around
every foreign call, so the checksum / indexing / normalisation work of the reader threads (datasets/ljspeech.py) runs in
parallel.  The library is plain C without a ROCm dependency; it is built by csrc/build.py (gcc) and, if missing, on first use."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsatt_io.so")

c_i64, c_u32, c_sz, _P = ctypes.c_int64, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_void_p

ERRORS = {-1: "truncated record header", -2: "corrupt length field", -3: "truncated record", -4: "corrupt record payload",
          -5: "too many entries for the caller's table", -6: "malformed protobuf message", -7: "bad argument",
          -8: "file cannot be opened or read"}


class ExampleFeature(ctypes.Structure):
    """satt_example_feature (include/satt_io.h)"""
    _fields_ = [("name_off", c_i64), ("name_len", c_i64), ("kind", ctypes.c_int32), ("packed", ctypes.c_int32),
                ("count", c_i64), ("val_off", c_i64), ("val_len", c_i64), ("first_int", c_i64)]


class Utterance(ctypes.Structure):
    """satt_utterance (include/satt_io.h)"""
    _fields_ = [(n, c_i64) for n in ("src_bytes", "tgt_bytes", "src_records", "tgt_records", "id", "source_length", "speaker_id",
                                     "age", "gender", "key_off", "key_len", "text_off", "text_len", "source_off", "source_count",
                                     "target_id", "target_length", "mel_width", "mel_off", "mel_count", "prepared_length")]


_SIGS = {
    "satt_io_version": (ctypes.c_int, []),
    "satt_io_last_errno": (ctypes.c_int, []),
    "satt_io_crc32c_hw": (ctypes.c_int, []),
    "satt_crc32c": (c_u32, [_P, c_sz]),
    "satt_crc32c_extend": (c_u32, [c_u32, _P, c_sz]),
    "satt_crc32c_sw": (c_u32, [_P, c_sz]),
    "satt_masked_crc32c": (c_u32, [_P, c_sz]),
    "satt_tfrecord_index": (c_i64, [_P, c_sz, ctypes.c_int, _P, _P, c_i64]),
    "satt_tfrecord_load": (c_i64, [ctypes.c_char_p, ctypes.c_int, _P, c_sz, _P, _P, _P, c_i64]),
    "satt_utterance_load": (c_i64, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, c_i64, _P, c_sz, ctypes.POINTER(Utterance)]),
    "satt_reader_create": (_P, [ctypes.c_int, ctypes.c_int, c_sz, ctypes.c_int, c_i64]),
    "satt_reader_submit": (c_i64, [_P, ctypes.c_char_p, ctypes.c_char_p]),
    "satt_reader_next": (c_i64, [_P, ctypes.POINTER(Utterance), ctypes.POINTER(_P), ctypes.POINTER(c_i64)]),
    "satt_reader_release": (ctypes.c_int, [_P, c_i64]),
    "satt_reader_outstanding": (c_i64, [_P]),
    "satt_reader_destroy": (None, [_P]),
    "satt_example_index": (c_i64, [_P, c_sz, ctypes.POINTER(ExampleFeature), c_i64]),
    "satt_example_int64s": (c_i64, [_P, c_sz, ctypes.c_int, _P, c_i64]),
    "satt_example_bytes": (c_i64, [_P, c_sz, _P, _P, c_i64]),
    "satt_prepare_mel": (c_i64, [_P, c_i64, c_i64, _P, c_i64, _P, c_i64, c_i64, ctypes.c_float, _P, c_i64]),
    "satt_prepared_length": (c_i64, [c_i64, c_i64]),
}

_lib = None


def _builder():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_satt_build", os.path.join(_HERE, "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def lib():
    """the loaded library; (re)built first when it is missing or older than csrc/host_io.c / include/satt_io.h (a stale .so used
    to surface as an AttributeError on the first new symbol).  The build is atomic and locked (csrc/build.py:build_io)."""
    global _lib
    if _lib is None:
        mod = _builder()
        if mod.io_stale():
            mod.build_io()
        l = ctypes.CDLL(_SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


_unavailable = None      # the reason of the first failed attempt (cached: a missing compiler does not come back within a process)


def available():
    """False when the library can neither be loaded nor built (no C compiler on this host): utils.tfrecord then falls back to its
    pure-Python checksum and framing code - slow, but a corpus can still be read and written.  The first failure is remembered
    (every crc32c() call used to re-run the lock and the compiler attempt) and reported ONCE."""
    global _unavailable
    if _lib is not None:
        return True
    if _unavailable is not None:
        return False
    try:
        lib()
        return True
    except (OSError, RuntimeError, AttributeError) as e:
        _unavailable = "%s: %s" % (type(e).__name__, e)
        import warnings
        warnings.warn("libsatt_io.so can neither be loaded nor built (%s): the TFRecord reader runs on its pure-Python path" % _unavailable,
                      RuntimeWarning, stacklevel=2)
        return False


def _addr(buf):
    """(address, length) of a bytes-like object without copying (bytes, bytearray, memoryview, contiguous numpy array)"""
    if isinstance(buf, np.ndarray):
        return buf.ctypes.data, buf.nbytes
    if isinstance(buf, bytes):
        return ctypes.cast(ctypes.c_char_p(buf), _P).value, len(buf)
    a = np.frombuffer(buf, dtype=np.uint8)
    return a.ctypes.data, a.nbytes


def crc32c(data):
    p, n = _addr(data)
    return int(lib().satt_crc32c(p, n))


def masked_crc32c(data):
    p, n = _addr(data)
    return int(lib().satt_masked_crc32c(p, n))


def tfrecord_index(buf, verify=True, max_records=None):
    """(offsets, lengths) int64 arrays of the record payloads inside a TFRecord file image; raises ValueError on damage"""
    p, n = _addr(buf)
    cap = max_records or 16
    while True:
        offs, lens = np.empty(cap, np.int64), np.empty(cap, np.int64)
        k = int(lib().satt_tfrecord_index(p, n, int(bool(verify)), offs.ctypes.data, lens.ctypes.data, cap))
        if k == -5 and not max_records:
            cap = max(4 * cap, n // 4096)
            continue
        if k < 0:
            raise ValueError(ERRORS.get(k, "error %d" % k))
        return offs[:k], lens[:k]


def tfrecord_load(path, verify=True, size_hint=1 << 19):
    """(file image as a uint8 array, offsets, lengths): open + read + framing / checksum check of one file in ONE foreign
    call (a reader thread gives up the GIL once per file)"""
    cap, rec = int(size_hint), 16
    nb = c_i64(0)
    bpath = os.fsencode(path)
    while True:
        buf = np.empty(cap, np.uint8)
        offs, lens = np.empty(rec, np.int64), np.empty(rec, np.int64)
        k = int(lib().satt_tfrecord_load(bpath, int(bool(verify)), buf.ctypes.data, cap, ctypes.byref(nb), offs.ctypes.data,
                                         lens.ctypes.data, rec))
        if k == -5:
            if nb.value > cap:
                cap = int(nb.value) + 1
            else:
                rec = max(4 * rec, cap // 4096)
            continue
        if k == -8:                       # SATT_IO_E_IO
            en = int(lib().satt_io_last_errno())
            raise OSError(en, os.strerror(en) if en else "cannot be opened or read", str(path))
        if k < 0:
            raise ValueError(ERRORS.get(k, "error %d" % k))
        return buf[:nb.value], offs[:k], lens[:k]


def utterance_load(source_path, target_path, r, verify=True, size_hint=400 << 10, arena=None):
    """(arena uint8 array, Utterance) of one `<key>.source.tfrecord` / `<key>.target.tfrecord` pair: both files read, checked
    and decoded in ONE foreign call (satt_utterance_load), i.e. one GIL release per utterance in a reader thread.
    arena: a uint8 array to read into (reused by the caller's buffer pool; replaced by a larger one when too small)"""
    cap = int(size_hint) if arena is None else arena.size
    u = Utterance()
    sp, tp = os.fsencode(source_path), os.fsencode(target_path)
    while True:
        if arena is None or arena.size < cap:
            arena = np.empty(cap, np.uint8)
        e = int(lib().satt_utterance_load(sp, tp, int(bool(verify)), int(r), arena.ctypes.data, cap, ctypes.byref(u)))
        if e == -5:
            cap = int(u.src_bytes + u.tgt_bytes) + 4096
            continue
        if e == -8:                       # SATT_IO_E_IO: the errno of the failing fopen / fread of THIS thread
            en = int(lib().satt_io_last_errno())
            bad = next((q for q in (source_path, target_path) if not os.path.exists(q)), source_path)
            raise OSError(en, os.strerror(en) if en else "cannot be opened or read", str(bad))
        if e == -7:
            raise ValueError("%s / %s: not an utterance record pair (a required feature is missing, source_length exceeds the "
                             "ids the record holds, or mel does not hold target_length x mel_width floats)" % (source_path, target_path))
        if e < 0:
            raise ValueError("%s / %s: %s" % (source_path, target_path, ERRORS.get(e, "error %d" % e)))
        return arena, u


class Reader:
    """satt_reader: `workers` POSIX threads load submitted (source, target) file pairs; next() delivers them in submission
    order.  The delivered arena is a view of a reader-owned buffer, leased until give_back(ticket); the C object is destroyed
    when the last Python reference (the pipeline's and those of undelivered leases) is gone."""

    def __init__(self, workers, slots, r, arena_bytes=400 << 10, verify=True):
        self._l = lib()
        self._h = self._l.satt_reader_create(int(workers), int(slots), int(arena_bytes), int(bool(verify)), int(r))
        if not self._h:
            raise MemoryError("satt_reader_create failed")
        self.slots = int(slots)

    def submit(self, source_path, target_path):
        """ticket, or None when the slot of the next ticket is still in use"""
        t = int(self._l.satt_reader_submit(self._h, os.fsencode(source_path), os.fsencode(target_path)))
        if t == -5:
            return None
        if t < 0:
            raise ValueError("satt_reader_submit: " + ERRORS.get(t, "error %d" % t))
        return t

    def outstanding(self):
        return int(self._l.satt_reader_outstanding(self._h))

    def next(self):
        """(ticket, status, Utterance, arena uint8 array) of the oldest outstanding ticket; blocks (GIL released) until loaded"""
        u, ptr, st = Utterance(), _P(), c_i64(0)
        t = int(self._l.satt_reader_next(self._h, ctypes.byref(u), ctypes.byref(ptr), ctypes.byref(st)))
        if t < 0:
            raise ValueError("satt_reader_next: nothing outstanding")
        arena = None
        if st.value == 0:
            n = int(u.src_bytes + u.tgt_bytes)
            arena = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n,))
        return t, int(st.value), u, arena

    def give_back(self, ticket):
        if self._h:
            self._l.satt_reader_release(self._h, int(ticket))

    def close(self):
        h, self._h = self._h, None
        if h:
            self._l.satt_reader_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_FEAT_CAP = 64


def example_index(payload):
    """{feature name: (kind, packed, count, val_off, val_len, first_int)} of a serialized tf.train.Example (offsets into payload)"""
    p, n = _addr(payload)
    feats = (ExampleFeature * _FEAT_CAP)()
    k = int(lib().satt_example_index(p, n, feats, _FEAT_CAP))
    if k < 0:
        raise ValueError(ERRORS.get(k, "error %d" % k))
    mv = memoryview(payload)
    out = {}
    for i in range(k):
        f = feats[i]
        out[bytes(mv[f.name_off:f.name_off + f.name_len]).decode("utf-8")] = (f.kind, f.packed, f.count, f.val_off, f.val_len,
                                                                              f.first_int)
    return out


def example_int64s(payload, entry):
    kind, packed, count, off, ln = entry[:5]
    if kind != 3:
        raise ValueError("not an int64_list feature")
    out = np.empty(max(1, count), np.int64)
    p, _ = _addr(payload)
    k = int(lib().satt_example_int64s(p + off, ln, packed, out.ctypes.data, out.size))
    if k < 0:
        raise ValueError(ERRORS.get(k, "error %d" % k))
    return out[:k]


def example_first_bytes(payload, entry):
    """zero-copy memoryview of the first value of a bytes_list feature"""
    kind, _, count, off, ln = entry[:5]
    if kind != 1 or count < 1:
        raise ValueError("not a non-empty bytes_list feature")
    return memoryview(payload)[off:off + ln]


def prepared_length(T, r):
    return int(lib().satt_prepared_length(int(T), int(r)))


def prepare_mel(mel, avg, std, r, silence, out):
    """out[rows_out, W] <- r silence frames | (mel - avg) / std | silence to the end; returns the prepared target length.
    mel: float32 [T, W] (C-contiguous), avg / std: float32 arrays with 1 or W entries, out: float32 [rows_out, W] C-contiguous."""
    T, W = mel.shape
    if mel.dtype != np.float32 or not mel.flags.c_contiguous or out.dtype != np.float32 or not out.flags.c_contiguous or \
            out.shape[1] != W:
        raise ValueError("prepare_mel: float32 C-contiguous [T, W] arrays of one width are required")
    L = int(lib().satt_prepare_mel(mel.ctypes.data, T, W, avg.ctypes.data, avg.size, std.ctypes.data, std.size, int(r),
                                   float(silence), out.ctypes.data, out.shape[0]))
    if L < 0:
        raise ValueError("prepare_mel: " + ERRORS.get(L, "error %d" % L) + " (rows_out too small, a stddev entry <= 0, or "
                         "average / stddev tables that have neither 1 nor %d entries)" % W)
    return L
