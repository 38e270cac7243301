"""Observability without TensorFlow / matplotlib:
* `EventFileWriter`: TensorBoard event files (`events.out.tfevents.*`) = TFRecord-framed `tensorflow.Event` protobufs
  carrying `Summary.Value{tag, simple_value}` scalars, with the reference's scalar names (models/models.py:600-616);
* `write_png` / `plot_alignments`: the `<key>.png` alignment plots of the reference's predict driver
  (predict_mel.py:58-66, modules/metrics.py plot_predictions) as plain grayscale / pseudo-colour PNGs (zlib only)."""
import os
import socket
import struct
import time
import zlib

import numpy as np

from .tfrecord import _enc_varint, _ld, masked_crc, parse_fields, read_records


def _f64(num, x):
    return _enc_varint((num << 3) | 1) + struct.pack("<d", float(x))


def _f32(num, x):
    return _enc_varint((num << 3) | 5) + struct.pack("<f", float(x))


def _vint(num, x):
    return _enc_varint((num << 3) | 0) + _enc_varint(int(x))


def encode_event(wall_time, step, scalars=None, file_version=None):
    """tensorflow.Event{wall_time=1 (double), step=2 (int64), file_version=3 (string) | summary=5 (Summary)};
    Summary{repeated Value value=1}; Value{tag=1 (string), simple_value=2 (float)}"""
    msg = _f64(1, wall_time) + _vint(2, step)
    if file_version is not None:
        msg += _ld(3, file_version.encode("utf-8"))
    if scalars:
        summary = b"".join(_ld(1, _ld(1, tag.encode("utf-8")) + _f32(2, val)) for tag, val in scalars.items())
        msg += _ld(5, summary)
    return msg


def decode_event(payload):
    """-> dict(wall_time, step, file_version | None, scalars {tag: value}) (what this writer emits)"""
    out = dict(wall_time=None, step=0, file_version=None, scalars={})
    for num, wt, v in parse_fields(payload):
        if num == 1 and wt == 1:
            out["wall_time"] = struct.unpack("<d", struct.pack("<Q", v))[0]
        elif num == 2 and wt == 0:
            out["step"] = v
        elif num == 3 and wt == 2:
            out["file_version"] = bytes(v).decode("utf-8")
        elif num == 5 and wt == 2:
            for n2, w2, val in parse_fields(v):
                if n2 != 1:
                    continue
                tag, sv = None, None
                for n3, w3, x in parse_fields(val):
                    if n3 == 1:
                        tag = bytes(x).decode("utf-8")
                    elif n3 == 2 and w3 == 5:
                        sv = struct.unpack("<f", struct.pack("<I", x))[0]
                if tag is not None:
                    out["scalars"][tag] = sv
    return out


class EventFileWriter:
    """append-only TensorBoard event file in `logdir` (the reference writes its summaries to the model directory)"""

    def __init__(self, logdir, filename_suffix=""):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s%s" % (int(time.time()), socket.gethostname(),
                                                                           filename_suffix))
        self._f = open(self.path, "ab")
        self._write(encode_event(time.time(), 0, file_version="brain.Event:2"))

    def _write(self, payload):
        hdr = struct.pack("<Q", len(payload))
        self._f.write(hdr + struct.pack("<I", masked_crc(hdr)) + payload + struct.pack("<I", masked_crc(payload)))

    def add_scalars(self, step, scalars, wall_time=None):
        self._write(encode_event(time.time() if wall_time is None else wall_time, step, scalars))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_events(path):
    return [decode_event(p) for p in read_records(path)]


# scalar names of the reference's summaries (models/models.py:600-616, get_validation_metrics :618-650)
TRAIN_SCALARS = ("mel_loss", "done_loss", "loss", "learning_rate")
EVAL_SCALARS = ("mel_loss", "done_loss", "loss_with_teacher", "mel_loss_with_teacher", "done_loss_with_teacher")


# ---------------------------------------------------------------------------------------------- PNG
def write_png(path, img):
    """img: uint8 [H, W] (grayscale) or [H, W, 3] (RGB)"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    color = 2 if img.ndim == 3 else 0
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(kind, data):
        c = struct.pack(">I", len(data)) + kind + data
        return c + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_png_size(path):
    with open(path, "rb") as f:
        d = f.read(33)
    assert d[:8] == b"\x89PNG\r\n\x1a\n" and d[12:16] == b"IHDR"
    w, h = struct.unpack(">II", d[16:24])
    return h, w


def _colormap(x):
    """x in [0,1] -> RGB (a simple perceptual ramp: dark blue -> green -> yellow)"""
    x = np.clip(x, 0.0, 1.0)
    r = np.clip(1.8 * x - 0.6, 0, 1); g = np.clip(1.4 * x, 0, 1) * (0.4 + 0.6 * x); b = np.clip(0.55 - 0.9 * np.abs(x - 0.25), 0, 1)
    return (np.stack([r, g, b], -1) * 255).astype(np.uint8)


def plot_alignments(path, alignments, scale=4, gap=6):
    """the alignment histories [T_memory, T_query] stacked vertically (memory axis up, as in the reference's plots)"""
    rows = []
    width = max(a.shape[1] for a in alignments) * scale
    for a in alignments:
        a = np.asarray(a, dtype=np.float32)
        img = _colormap(a / max(float(a.max()), 1e-12))[::-1]                 # memory index grows upwards
        img = np.repeat(np.repeat(img, scale, axis=0), scale, axis=1)
        pad = np.zeros((img.shape[0], width - img.shape[1], 3), np.uint8)
        rows.append(np.concatenate([img, pad], axis=1))
        rows.append(np.full((gap, width, 3), 255, np.uint8))
    write_png(path, np.concatenate(rows[:-1], axis=0))
