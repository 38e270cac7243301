"""The protobuf layers of the TensorFlow file formats this repo reads and writes without TensorFlow - tf.train.Example (corpus records:
reference utils/tfrecord.py:82-152), tensorflow.Event / Summary (TensorBoard scalars: reference models/models.py:600-616) and
BundleHeaderProto / BundleEntryProto (checkpoint index: reference train.py:76-78) - cross-checked against an INDEPENDENT implementation
of the wire format: Google's `protobuf` runtime, with the message schemas restated here from the published .proto files
(tensorflow/core/example/{example,feature}.proto, core/util/event.proto, core/framework/{summary,tensor_shape,versions}.proto,
core/protobuf/tensor_bundle.proto).  Until r6 these formats were verified by self round trips only; a
TensorFlow-written file still cannot be had in this container, so the table / block layer of the checkpoint index stays self-checked."""
import os
import struct

import numpy as np
import pytest

pb = pytest.importorskip("google.protobuf")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

F = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None, packed=None, oneof=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if packed is not None:
        f.options.packed = packed
    if oneof is not None:
        f.oneof_index = oneof
    return f


def _schema():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "satt_tf_formats.proto", "tfx", "proto3"
    # ---- feature.proto / example.proto
    m = fd.message_type.add(); m.name = "BytesList"; _field(m, "value", 1, F.TYPE_BYTES, F.LABEL_REPEATED)
    m = fd.message_type.add(); m.name = "FloatList"; _field(m, "value", 1, F.TYPE_FLOAT, F.LABEL_REPEATED, packed=True)
    m = fd.message_type.add(); m.name = "Int64List"; _field(m, "value", 1, F.TYPE_INT64, F.LABEL_REPEATED, packed=True)
    m = fd.message_type.add(); m.name = "Feature"
    m.oneof_decl.add().name = "kind"
    _field(m, "bytes_list", 1, F.TYPE_MESSAGE, type_name=".tfx.BytesList", oneof=0)
    _field(m, "float_list", 2, F.TYPE_MESSAGE, type_name=".tfx.FloatList", oneof=0)
    _field(m, "int64_list", 3, F.TYPE_MESSAGE, type_name=".tfx.Int64List", oneof=0)
    m = fd.message_type.add(); m.name = "Features"
    e = m.nested_type.add(); e.name = "FeatureEntry"; e.options.map_entry = True
    _field(e, "key", 1, F.TYPE_STRING); _field(e, "value", 2, F.TYPE_MESSAGE, type_name=".tfx.Feature")
    _field(m, "feature", 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, type_name=".tfx.Features.FeatureEntry")
    m = fd.message_type.add(); m.name = "Example"; _field(m, "features", 1, F.TYPE_MESSAGE, type_name=".tfx.Features")
    # ---- summary.proto / event.proto (the fields this repo uses; the oneofs as in the published files)
    m = fd.message_type.add(); m.name = "SummaryValue"
    m.oneof_decl.add().name = "value"
    _field(m, "tag", 1, F.TYPE_STRING); _field(m, "node_name", 7, F.TYPE_STRING)
    _field(m, "simple_value", 2, F.TYPE_FLOAT, oneof=0)
    m = fd.message_type.add(); m.name = "Summary"; _field(m, "value", 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, type_name=".tfx.SummaryValue")
    m = fd.message_type.add(); m.name = "Event"
    m.oneof_decl.add().name = "what"
    _field(m, "wall_time", 1, F.TYPE_DOUBLE); _field(m, "step", 2, F.TYPE_INT64)
    _field(m, "file_version", 3, F.TYPE_STRING, oneof=0); _field(m, "graph_def", 4, F.TYPE_BYTES, oneof=0)
    _field(m, "summary", 5, F.TYPE_MESSAGE, type_name=".tfx.Summary", oneof=0)
    # ---- tensor_shape.proto / versions.proto / tensor_bundle.proto
    m = fd.message_type.add(); m.name = "TensorShapeProto"
    d = m.nested_type.add(); d.name = "Dim"; _field(d, "size", 1, F.TYPE_INT64); _field(d, "name", 2, F.TYPE_STRING)
    _field(m, "dim", 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, type_name=".tfx.TensorShapeProto.Dim"); _field(m, "unknown_rank", 3, F.TYPE_BOOL)
    m = fd.message_type.add(); m.name = "VersionDef"
    _field(m, "producer", 1, F.TYPE_INT32); _field(m, "min_consumer", 2, F.TYPE_INT32)
    _field(m, "bad_consumers", 3, F.TYPE_INT32, F.LABEL_REPEATED, packed=True)
    m = fd.message_type.add(); m.name = "BundleHeaderProto"
    _field(m, "num_shards", 1, F.TYPE_INT32); _field(m, "endianness", 2, F.TYPE_INT32)      # (enum LITTLE = 0, BIG = 1: int32 on the wire)
    _field(m, "version", 3, F.TYPE_MESSAGE, type_name=".tfx.VersionDef")
    m = fd.message_type.add(); m.name = "BundleEntryProto"
    _field(m, "dtype", 1, F.TYPE_INT32)                                                     # (enum DataType: DT_FLOAT = 1, DT_INT32 = 3, DT_INT64 = 9 ...)
    _field(m, "shape", 2, F.TYPE_MESSAGE, type_name=".tfx.TensorShapeProto"); _field(m, "shard_id", 3, F.TYPE_INT32)
    _field(m, "offset", 4, F.TYPE_INT64); _field(m, "size", 5, F.TYPE_INT64); _field(m, "crc32c", 6, F.TYPE_FIXED32)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("tfx." + n))
    return {n: get(n) for n in ("Example", "Feature", "Event", "Summary", "BundleHeaderProto", "BundleEntryProto")}


M = _schema()


def test_example_records_are_what_an_independent_protobuf_reads_and_writes(satt):
    from satt_amd.utils import tfrecord
    g = np.random.default_rng(0)
    mel = g.normal(-40, 10, (37, 80)).astype("<f4")
    src = np.concatenate([[0], g.integers(1, 60, 11), [0]]).astype("<i8")
    feats = {"id": 7, "key": b"LJ001-0001", "source": src.tobytes(), "source_length": 13, "text": "grüß".encode("utf-8"),
             "mel": mel.tobytes(), "mel_width": 80, "target_length": 37}
    ours = tfrecord.make_example(feats)
    ex = M["Example"]()
    ex.ParseFromString(ours)                                  # Google's decoder on OUR bytes
    got = ex.features.feature
    assert set(got) == set(feats)
    for k, v in feats.items():
        if isinstance(v, bytes):
            assert got[k].WhichOneof("kind") == "bytes_list" and list(got[k].bytes_list.value) == [v]
        else:
            assert got[k].WhichOneof("kind") == "int64_list" and list(got[k].int64_list.value) == [v]
    # the other direction: an Example serialised by Google's encoder (its own field / map order, packed int64 and float lists)
    ex2 = M["Example"]()
    for k, v in feats.items():
        if isinstance(v, bytes):
            ex2.features.feature[k].bytes_list.value.append(v)
        else:
            ex2.features.feature[k].int64_list.value.append(v)
    ex2.features.feature["neg"].int64_list.value.extend([-1, -(1 << 40), 5])
    ex2.features.feature["floats"].float_list.value.extend([1.5, -2.25, 3.0])
    theirs = ex2.SerializeToString()
    back = tfrecord.parse_example(theirs)
    for k, v in feats.items():
        r = back[k]
        assert (bytes(r[0]) == v and len(r) == 1) if isinstance(v, bytes) else (list(r) == [v]), k
    assert list(back["neg"]) == [-1, -(1 << 40), 5] and np.allclose(np.asarray(back["floats"], dtype=np.float64), [1.5, -2.25, 3.0])
    # and the native library (csrc/host_io.c: satt_example_index / satt_example_int64s) indexes Google's bytes the same way
    from satt_amd import _io
    if _io.available():
        idx = _io.example_index(theirs)
        assert set(idx) == set(feats) | {"neg", "floats"}
        assert idx["source"][0] == 1 and idx["neg"][0] == 3 and idx["floats"][0] == 2          # kinds: bytes / int64 / float lists
        assert list(_io.example_int64s(theirs, idx["neg"])) == [-1, -(1 << 40), 5]
        assert idx["target_length"][5] == 37 and idx["id"][5] == 7                               # first_int of the scalar features


def test_event_file_records_against_an_independent_protobuf(satt, tmp_path):
    from satt_amd.utils import summary, tfrecord
    ours = summary.encode_event(1234567.875, 42, scalars={"loss": 1.5, "mel_loss": 0.25, "learning_rate": 5e-4})
    ev = M["Event"]()
    ev.ParseFromString(ours)
    assert ev.wall_time == 1234567.875 and ev.step == 42 and ev.WhichOneof("what") == "summary"
    assert {v.tag: v.simple_value for v in ev.summary.value} == {"loss": 1.5, "mel_loss": 0.25, "learning_rate": np.float32(5e-4)}
    first = M["Event"](); first.ParseFromString(summary.encode_event(1.0, 0, file_version="brain.Event:2"))
    assert first.WhichOneof("what") == "file_version" and first.file_version == "brain.Event:2"
    # Google-serialised events, framed as TFRecords by our writer, through our reader
    ev2 = M["Event"](); ev2.wall_time = 99.5; ev2.step = (1 << 40) + 3
    v = ev2.summary.value.add(); v.tag = "done_loss"; v.simple_value = 0.125
    p = str(tmp_path / "events.out.tfevents.1.host")
    tfrecord.write_records(p, [first.SerializeToString(), ev2.SerializeToString()])
    got = summary.read_events(p)
    assert got[0]["file_version"] == "brain.Event:2" and got[1]["step"] == (1 << 40) + 3 and got[1]["wall_time"] == 99.5
    assert got[1]["scalars"] == {"done_loss": 0.125}
    # and the file our writer produces parses record by record with Google's decoder
    w = summary.EventFileWriter(str(tmp_path / "run"))
    w.add_scalars(7, {"loss": 2.0}); w.flush()
    f = [os.path.join(tmp_path / "run", n) for n in os.listdir(tmp_path / "run")][0]
    recs = [bytes(r) for r in tfrecord.read_records(f)]
    evs = []
    for r in recs:
        e = M["Event"](); e.ParseFromString(r); evs.append(e)
    assert evs[0].file_version.startswith("brain.Event:") and evs[-1].step == 7 and evs[-1].summary.value[0].tag == "loss"


def test_checkpoint_index_entries_against_an_independent_protobuf(satt, tmp_path):
    from satt_amd.utils import tf_checkpoint as ck
    g = np.random.default_rng(1)
    tensors = {"model/dense/kernel": g.normal(size=(5, 3)).astype(np.float32), "model/step": np.array(12345, dtype=np.int64),
               "model/ids": np.arange(7, dtype=np.int32), "model/empty_dim": np.zeros((2, 0, 4), dtype=np.float32)}
    prefix = str(tmp_path / "model.ckpt-1")
    ck.write_checkpoint(prefix, tensors)
    entries = dict(ck.read_table(prefix + ".index"))
    hdr = M["BundleHeaderProto"](); hdr.ParseFromString(bytes(entries[b""]))
    assert hdr.num_shards == 1 and hdr.endianness == 0 and hdr.version.producer == 1
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    dt = {np.dtype(np.float32): 1, np.dtype(np.int32): 3, np.dtype(np.int64): 9}
    for name, a in tensors.items():
        e = M["BundleEntryProto"](); e.ParseFromString(bytes(entries[name.encode()]))      # Google's decoder on OUR entry
        assert e.dtype == dt[a.dtype] and [d.size for d in e.shape.dim] == list(a.shape) and e.shard_id == 0 and e.size == a.nbytes
        raw = data[e.offset:e.offset + e.size]
        assert np.array_equal(np.frombuffer(raw, dtype=a.dtype).reshape(a.shape), a)
        assert e.crc32c == ck.masked_crc(raw)
        # ... and OUR entry parser on Google's serialisation of the same message
        back = ck._parse_entry(e.SerializeToString())
        assert (back["dtype"], back["shape"], back["offset"], back["size"], back["crc32c"]) == (e.dtype, list(a.shape), e.offset, e.size, e.crc32c)
    # a checkpoint whose index VALUES are Google-serialised protos (table layer ours) reads back through CheckpointReader
    items, blob = [], bytearray()
    for name in sorted(tensors, key=lambda s: s.encode()):
        a = tensors[name]; raw = a.tobytes()
        e = M["BundleEntryProto"](); e.dtype = dt[a.dtype]; e.offset = len(blob); e.size = len(raw); e.crc32c = ck.masked_crc(raw)
        for d in a.shape:
            e.shape.dim.add().size = d
        items.append((name.encode(), e.SerializeToString())); blob += raw
    h = M["BundleHeaderProto"](); h.num_shards = 1; h.version.producer = 1
    p2 = str(tmp_path / "model.ckpt-2")
    ck.write_table(p2 + ".index", [(b"", h.SerializeToString())] + items)
    open(p2 + ".data-00000-of-00001", "wb").write(bytes(blob))
    r = ck.CheckpointReader(p2)
    for name, a in tensors.items():
        assert np.array_equal(r.get_tensor(name), a) and r.get_tensor(name).dtype == a.dtype, name
    assert struct.unpack("<Q", open(p2 + ".index", "rb").read()[-8:])[0] == 0xdb4775248b80fb57      # the table magic (leveldb)
