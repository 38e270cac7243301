"""TensorFlow checkpoint ("tensor bundle", `<prefix>.index` + `<prefix>.data-00000-of-0000N`) reader and writer without
TensorFlow - the file side of the reference's warm start (reference train.py:76-78: tf.estimator.WarmStartSettings(
ckpt_to_initialize_from=..., vars_to_warm_start=...), hparams.py:200-202) and of exchanging weights with it.

Format (TensorFlow core/util/tensor_bundle + core/lib/io/table, which is LevelDB's table format):
  * `.index` is a sorted string table.  Data blocks hold entries `varint32 shared | varint32 non_shared | varint32 value_len |
    key suffix | value` (keys prefix-compressed against the previous key, restart points every 16 entries), followed by the
    uint32 restart offsets and their count; every block carries a 5-byte trailer: compression type (0 = none) + masked
    CRC-32C of block and type byte.  An index block maps the last key of each data block to its (offset, size) handle; the
    48-byte footer holds the metaindex and index handles (varint64 pairs, zero padded to 40 bytes) and the magic
    0xdb4775248b80fb57.
  * key "" -> BundleHeaderProto {1: num_shards, 2: endianness (0 = little), 3: VersionDef {1: producer}};
    key <variable name> -> BundleEntryProto {1: dtype, 2: TensorShapeProto {2: dim {1: size}}, 3: shard_id, 4: offset, 5: size,
    6: fixed32 masked CRC-32C of the tensor bytes}.
  * `.data-*` are the raw little-endian tensor bytes at those offsets.
Snappy-compressed index blocks (type 1) are refused: TensorFlow's BundleWriter writes uncompressed tables."""
import os
import struct

import numpy as np

from .tfrecord import _TABLE, _enc_varint, _ld, _varint, parse_fields

MAGIC = 0xdb4775248b80fb57
# tensorflow/core/framework/types.proto
DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"), 6: np.dtype("i1"),
          9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"), 22: np.dtype("<u4"), 23: np.dtype("<u8")}
DTYPE_IDS = {v: k for k, v in DTYPES.items()}


class CheckpointError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------- CRC-32C of large buffers
def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1; i += 1
    return s


def _gf2_square(mat):
    return [_gf2_times(mat, mat[n]) for n in range(32)]


def _zeros_operator(nbytes):
    """the GF(2) matrix that advances a CRC-32C register over `nbytes` zero bytes"""
    odd = [0x82F63B78] + [1 << n for n in range(31)]        # one zero bit
    even = _gf2_square(odd)                                 # two
    odd = _gf2_square(even)                                 # four
    n = nbytes
    mats = []
    while n:                                                # even = 8 bits (one byte) after the first squaring below
        even = _gf2_square(odd)
        if n & 1:
            mats.append(even)
        n >>= 1
        if not n:
            break
        odd = _gf2_square(even)
        if n & 1:
            mats.append(odd)
        n >>= 1
    return mats


def _apply(mats, crc):
    for m in mats:
        crc = _gf2_times(m, crc)
    return crc


def crc32c(data):
    """CRC-32C of a bytes-like object.  Large inputs are cut into equal blocks whose CRCs are computed side by side with
    numpy (one table step per byte POSITION for all blocks at once) and then combined with the zero-extension operator."""
    buf = np.frombuffer(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview, np.ndarray)) else data, dtype=np.uint8)
    n = buf.size
    if n == 0:
        return 0
    nblk = 1 if n < 1 << 14 else min(8192, n // 1024)
    L = n // nblk
    head = buf[:nblk * L].reshape(nblk, L)
    crc = np.full(nblk, 0xFFFFFFFF, dtype=np.uint32)
    for i in range(L):
        crc = _TABLE[(crc ^ head[:, i]) & 0xFF] ^ (crc >> 8)
    crc ^= 0xFFFFFFFF
    total = int(crc[0])
    if nblk > 1:
        op = _zeros_operator(L)
        for k in range(1, nblk):
            total = _apply(op, total) ^ int(crc[k])
    tail = buf[nblk * L:]
    if tail.size:
        c = 0xFFFFFFFF
        for b in tail.tolist():
            c = int(_TABLE[(c ^ b) & 0xFF]) ^ (c >> 8)
        total = _apply(_zeros_operator(int(tail.size)), total) ^ (c ^ 0xFFFFFFFF)
    return total


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------- string table (.index)
def _read_block(buf, offset, size, verify):
    raw = buf[offset:offset + size + 5]
    if len(raw) != size + 5:
        raise CheckpointError("index file truncated (block at %d)" % offset)
    if raw[size] != 0:
        raise CheckpointError("compressed index block (type %d): only uncompressed tables are supported" % raw[size])
    if verify and struct.unpack_from("<I", raw, size + 1)[0] != masked_crc(raw[:size + 1]):
        raise CheckpointError("index block checksum mismatch at offset %d" % offset)
    return memoryview(raw)[:size]


def _block_entries(block):
    n = len(block)
    nrestart = struct.unpack_from("<I", block, n - 4)[0]
    end = n - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _handle(buf, pos):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def read_table(path, verify=True):
    """[(key bytes, value memoryview)] of a LevelDB-format table file, in key order"""
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != MAGIC:
        raise CheckpointError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    foot = memoryview(buf)[len(buf) - 48:]
    _, _, p = _handle(foot, 0)                      # metaindex handle (unused)
    ioff, isize, _ = _handle(foot, p)
    out = []
    for _, hv in _block_entries(_read_block(buf, ioff, isize, verify)):
        off, size, _ = _handle(hv, 0)
        out += list(_block_entries(_read_block(buf, off, size, verify)))
    return out


def _build_block(entries, restart_interval=16):
    out, restarts, prev, cnt = bytearray(), [], b"", 0
    for key, val in entries:
        shared = 0
        if cnt % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(prev), len(key))
            while shared < m and prev[shared] == key[shared]:
                shared += 1
        out += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(val)) + key[shared:] + bytes(val)
        prev = key; cnt += 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path, entries, block_size=4096):
    """entries: [(key bytes, value bytes)] sorted by key -> LevelDB-format table (uncompressed blocks)"""
    f = bytearray()
    index, cur, cur_bytes = [], [], 0

    def emit(block):
        off = len(f)
        f.extend(block); f.append(0)
        f.extend(struct.pack("<I", masked_crc(block + b"\0")))
        return off, len(block)

    def flush():
        nonlocal cur, cur_bytes
        if cur:
            off, size = emit(_build_block(cur))
            index.append((cur[-1][0], _enc_varint(off) + _enc_varint(size)))
            cur, cur_bytes = [], 0
    for key, val in entries:
        cur.append((key, val)); cur_bytes += len(key) + len(val) + 3
        if cur_bytes >= block_size:
            flush()
    flush()
    moff, msize = emit(_build_block([]))                    # empty metaindex block
    ioff, isize = emit(_build_block(index, restart_interval=1))
    foot = _enc_varint(moff) + _enc_varint(msize) + _enc_varint(ioff) + _enc_varint(isize)
    f.extend(foot + b"\0" * (40 - len(foot)) + struct.pack("<Q", MAGIC))
    open(path, "wb").write(bytes(f))


# ---------------------------------------------------------------------------------------------- tensor bundle
def _parse_entry(val):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for num, wt, v in parse_fields(val):
        if num == 1:
            e["dtype"] = v
        elif num == 2:
            for n2, _, v2 in parse_fields(v):
                if n2 == 2:
                    size = 0
                    for n3, _, v3 in parse_fields(v2):
                        if n3 == 1:
                            size = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                    e["shape"].append(size)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["sliced"] = True
    return e


class CheckpointReader:
    """`tf.train.load_checkpoint`-like view of `<prefix>.index` / `<prefix>.data-*`."""

    def __init__(self, prefix, verify=True):
        self.prefix, self.verify = prefix, verify
        if not os.path.exists(prefix + ".index"):
            raise CheckpointError("no checkpoint index at %s.index" % prefix)
        self.entries, self.num_shards = {}, 1
        for key, val in read_table(prefix + ".index", verify):
            if key == b"":
                for num, _, v in parse_fields(val):
                    if num == 1:
                        self.num_shards = v
                    elif num == 2 and v != 0:
                        raise CheckpointError("big-endian checkpoint")
            else:
                self.entries[key.decode()] = _parse_entry(val)
        self._data = {}

    def get_variable_to_shape_map(self):
        return {k: list(e["shape"]) for k, e in self.entries.items()}

    def has_tensor(self, name):
        return name in self.entries

    def get_tensor(self, name):
        e = self.entries.get(name)
        if e is None:
            raise CheckpointError("variable %r is not in the checkpoint %s" % (name, self.prefix))
        if e["sliced"]:
            raise CheckpointError("variable %r is stored in slices (partitioned variable): not supported" % name)
        if e["dtype"] not in DTYPES:
            raise CheckpointError("variable %r has unsupported dtype id %d" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in self._data:
            path = "%s.data-%05d-of-%05d" % (self.prefix, sid, self.num_shards)
            self._data[sid] = np.memmap(path, dtype=np.uint8, mode="r")
        raw = self._data[sid][e["offset"]:e["offset"] + e["size"]]
        dt = DTYPES[e["dtype"]]
        n = int(np.prod(e["shape"])) if e["shape"] else 1
        if raw.size != n * dt.itemsize:
            raise CheckpointError("variable %r: %d bytes on disk, shape %s needs %d" % (name, raw.size, e["shape"], n * dt.itemsize))
        if self.verify and e["crc32c"] is not None and masked_crc(raw) != e["crc32c"]:
            raise CheckpointError("variable %r: data checksum mismatch" % name)
        return np.frombuffer(bytes(raw), dtype=dt).reshape(e["shape"]).copy()


def write_checkpoint(prefix, tensors):
    """{variable name: array} -> `<prefix>.index` + `<prefix>.data-00000-of-00001` (one shard, little-endian)"""
    data = bytearray()
    items = []
    for name in sorted(tensors, key=lambda s: s.encode()):
        a = np.asarray(tensors[name], order="C")        # (ascontiguousarray would promote scalars to shape [1])
        if a.dtype.byteorder == ">":
            a = a.astype(a.dtype.newbyteorder("<"))
        did = DTYPE_IDS.get(a.dtype)
        if did is None:
            raise CheckpointError("dtype %s of %r has no TensorFlow checkpoint id here" % (a.dtype, name))
        raw = a.tobytes()
        shape = b"".join(_ld(2, _enc_varint(8) + _enc_varint(int(d))) for d in a.shape)
        entry = _enc_varint(8) + _enc_varint(did) + _ld(2, shape) + _enc_varint(32) + _enc_varint(len(data)) + \
            _enc_varint(40) + _enc_varint(len(raw)) + _enc_varint((6 << 3) | 5) + struct.pack("<I", masked_crc(raw))
        items.append((name.encode(), entry))
        data += raw
    header = _enc_varint(8) + _enc_varint(1) + _ld(3, _enc_varint(8) + _enc_varint(1))      # num_shards 1, producer 1
    write_table(prefix + ".index", [(b"", header)] + items)
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
