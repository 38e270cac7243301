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


def crc32c(data):
    """CRC-32C (Castagnoli), table driven."""
    crc = 0xFFFFFFFF
    tab = _TABLE
    for b in bytes(data):
        crc = int(tab[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


class TFRecordError(ValueError):
    pass


def read_records(path, verify=True):
    """yield the payload bytes of every record of a TFRecord file"""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise TFRecordError("%s: truncated record header" % path)
            n, hcrc = struct.unpack("<QI", head)
            if verify and masked_crc(head[:8]) != hcrc:
                raise TFRecordError("%s: corrupt length field" % path)
            data = f.read(n)
            tail = f.read(4)
            if len(data) < n or len(tail) < 4:
                raise TFRecordError("%s: truncated record" % path)
            if verify and masked_crc(data) != struct.unpack("<I", tail)[0]:
                raise TFRecordError("%s: corrupt record payload" % path)
            yield data


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
