"""TFRecord files and tf.train.Example payloads without TensorFlow.

The reference stores one record per utterance in `<key>.source.tfrecord` / `<key>.target.tfrecord`
(reference utils/tfrecord.py:43-104, datasets/ljspeech/dataset.py:52-72).  Formats restated from their published
definitions:
  * TFRecord framing: uint64 length | uint32 masked_crc32c(length) | payload | uint32 masked_crc32c(payload)
    (little endian; crc32c = Castagnoli polynomial; mask = rotr15(crc) + 0xa282ead8);
  * payload: protobuf `Example{ Features features = 1 }`, `Features{ map<string, Feature> feature = 1 }`,
    `Feature{ oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3 } }`,
    each list `{ repeated value = 1 }` (floats / varints packed or unpacked).
Only what the dataset needs is implemented: a reader, a writer (for tests and for producing fixtures) and a tiny
wire-format codec."""
import struct

import numpy as np

_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)
_TABLE = np.array(_TABLE, dtype=np.uint32)


def crc32c_py(data):
    """CRC-32C (Castagnoli), table driven, byte at a time: the plain restatement the C routine is tested against (~3 MB/s)."""
    crc = 0xFFFFFFFF
    tab = _TABLE
    for b in bytes(data):
        crc = int(tab[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def crc32c(data):
    """CRC-32C through libsatt_io.so (include/satt_io.h: SSE4.2 crc32 instruction, slicing-by-8 tables without it); the pure-Python
    table walk when the library can neither be loaded nor built on this host"""
    from .. import _io
    if not _io.available():
        return crc32c_py(data)
    return _io.crc32c(data)


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


class TFRecordError(ValueError):
    pass


def _read_record_views_py(path, verify):
    """the framing walk in Python (fallback without libsatt_io.so): same checks, same error class"""
    buf = open(path, "rb").read()
    mv, pos, out = memoryview(buf), 0, []
    while pos < len(buf):
        if pos + 12 > len(buf):
            raise TFRecordError("%s: truncated record header" % path)
        n, = struct.unpack_from("<Q", buf, pos)
        if verify and struct.unpack_from("<I", buf, pos + 8)[0] != masked_crc(buf[pos:pos + 8]):
            raise TFRecordError("%s: corrupt length field" % path)
        if pos + 12 + n + 4 > len(buf):
            raise TFRecordError("%s: truncated record" % path)
        if verify and struct.unpack_from("<I", buf, pos + 12 + n)[0] != masked_crc(mv[pos + 12:pos + 12 + n]):
            raise TFRecordError("%s: corrupt record payload" % path)
        out.append(mv[pos + 12:pos + 12 + n])
        pos += 16 + n
    return out


def read_record_views(path, verify=True):
    """zero-copy memoryviews of the payloads of every record of a TFRecord file (one read of the file; framing and both
    checksums of every record checked in C: satt_tfrecord_index)"""
    from .. import _io
    if not _io.available():
        return _read_record_views_py(path, verify)
    try:
        buf, offs, lens = _io.tfrecord_load(path, verify)
    except ValueError as e:
        raise TFRecordError("%s: %s" % (path, e)) from None
    mv = memoryview(buf)
    return [mv[o:o + n] for o, n in zip(offs.tolist(), lens.tolist())]


def read_records(path, verify=True):
    """yield the payload bytes of every record of a TFRecord file"""
    for v in read_record_views(path, verify):
        yield bytes(v)


def write_records(path, payloads):
    with open(path, "wb") as f:
        for p in payloads:
            head = struct.pack("<Q", len(p))
            f.write(head + struct.pack("<I", masked_crc(head)) + p + struct.pack("<I", masked_crc(p)))


# ---- protobuf wire format (varint / length-delimited / fixed32 only) --------------------------------------------
def _varint(buf, pos):
    x, shift = 0, 0
    while True:
        b = buf[pos]; pos += 1
        x |= (b & 0x7F) << shift
        if not b & 0x80:
            return x, pos
        shift += 7


def _fields(buf):
    """iterate (field number, wire type, value) of one message; value: int (varint/fixed) or memoryview (bytes)"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        else:
            raise TFRecordError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


def parse_fields(buf):
    """(field number, wire type, value) triples of one protobuf message (bytes or memoryview)"""
    return _fields(memoryview(buf) if not isinstance(buf, memoryview) else buf)


def _to_signed(x):
    return x - (1 << 64) if x >= (1 << 63) else x


def parse_example(payload):
    """tf.train.Example -> {name: list of bytes | np.float32 array | np.int64 array}"""
    buf = memoryview(payload)
    out = {}
    for num, wt, feats in _fields(buf):
        if num != 1 or wt != 2:
            continue
        for fnum, fwt, entry in _fields(feats):                 # map entries
            if fnum != 1 or fwt != 2:
                continue
            name, feature = None, None
            for enum_, ewt, v in _fields(entry):
                if enum_ == 1:
                    name = bytes(v).decode("utf-8")
                elif enum_ == 2:
                    feature = v
            if name is None or feature is None:
                continue
            value = []
            for knum, kwt, lst in _fields(feature):
                if knum == 1:                                    # BytesList
                    value = [bytes(v) for n2, w2, v in _fields(lst) if n2 == 1]
                elif knum == 2:                                  # FloatList (packed or not)
                    vals = []
                    for n2, w2, v in _fields(lst):
                        if n2 != 1:
                            continue
                        if w2 == 2:
                            vals.append(np.frombuffer(bytes(v), dtype="<f4"))
                        else:
                            vals.append(np.array([struct.unpack("<f", struct.pack("<I", v))[0]], dtype=np.float32))
                    value = np.concatenate(vals) if vals else np.zeros(0, np.float32)
                elif knum == 3:                                  # Int64List (packed or not)
                    vals = []
                    for n2, w2, v in _fields(lst):
                        if n2 != 1:
                            continue
                        if w2 == 2:
                            p, mv = 0, v
                            while p < len(mv):
                                x, p = _varint(mv, p)
                                vals.append(_to_signed(x))
                        else:
                            vals.append(_to_signed(v))
                    value = np.array(vals, dtype=np.int64)
            out[name] = value
    return out


def _enc_varint(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def make_example(features):
    """{name: bytes | list of bytes | int / ints | float / floats (np.floating arrays)} -> serialized tf.train.Example"""
    entries = b""
    for name in sorted(features):
        v = features[name]
        if isinstance(v, (bytes, bytearray)):
            v = [bytes(v)]
        if isinstance(v, (list, tuple)) and v and isinstance(v[0], (bytes, bytearray)):
            feat = _ld(1, b"".join(_ld(1, bytes(x)) for x in v))
        else:
            a = np.atleast_1d(np.asarray(v))
            if np.issubdtype(a.dtype, np.floating):
                feat = _ld(2, _ld(1, a.astype("<f4").tobytes()))
            else:
                feat = _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in a)))
        entries += _ld(1, _ld(1, name.encode("utf-8")) + _ld(2, feat))
    return _ld(1, entries)


def write_prediction_result(id_, key, alignments, mel, ground_truth_mel, text, source, accent_type, filename):
    """One prediction as a single-record TFRecord file with the reference's feature names and encodings
    (reference utils/tfrecord.py:135-152): arrays as raw little-endian bytes, alignments as a bytes list."""
    mel = np.ascontiguousarray(mel, dtype="<f4")
    gt = np.ascontiguousarray(ground_truth_mel if ground_truth_mel is not None else np.zeros((0, mel.shape[1])), dtype="<f4")
    src = np.ascontiguousarray(source, dtype="<i8")
    feats = {
        "id": [int(id_)],
        "key": [key.encode("utf-8")],
        "mel": [mel.tobytes()],
        "mel_length": [mel.shape[0]],
        "mel_width": [mel.shape[1]],
        "ground_truth_mel": [gt.tobytes()],
        "ground_truth_mel_length": [gt.shape[0]],
        "alignment": [np.ascontiguousarray(a, dtype="<f4").tobytes() for a in alignments],
        "text": [text.encode("utf-8")],
        "source": [src.tobytes()],
        "source_length": [src.shape[0]],
        "accent_type": [np.ascontiguousarray(accent_type).tobytes()] if accent_type is not None else [],
    }
    # an empty bytes list has to be spelled out (make_example infers the kind from the first element)
    payload = make_example({k: v for k, v in feats.items() if v != []})
    if feats["accent_type"] == []:
        payload = _append_empty_bytes_feature(payload, "accent_type")
    write_records(filename, [payload])


def _append_empty_bytes_feature(example, name):
    """add `name: Feature{bytes_list{}}` to a serialized Example (its only field is the Features message)"""
    fields = list(_fields(memoryview(example)))
    assert len(fields) == 1 and fields[0][0] == 1
    entries = bytes(fields[0][2]) + _ld(1, _ld(1, name.encode("utf-8")) + _ld(2, _ld(1, b"")))
    return _ld(1, entries)


def parse_prediction_result(payload):
    """inverse of write_prediction_result: arrays restored to their shapes"""
    f = parse_example(payload)
    w = int(f["mel_width"][0])
    sl = int(f["source_length"][0])
    n_align = len(f["alignment"])
    out = dict(id=int(f["id"][0]), key=f["key"][0].decode("utf-8"), text=f["text"][0].decode("utf-8"),
               mel=np.frombuffer(f["mel"][0], "<f4").reshape(int(f["mel_length"][0]), w),
               ground_truth_mel=np.frombuffer(f["ground_truth_mel"][0], "<f4").reshape(int(f["ground_truth_mel_length"][0]), w),
               source=np.frombuffer(f["source"][0], "<i8"),
               alignment=[np.frombuffer(a, "<f4").reshape(sl, -1) for a in f["alignment"]] if n_align else [])
    return out
