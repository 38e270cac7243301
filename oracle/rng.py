"""TEST INFRASTRUCTURE ONLY (oracle). Counter-based dropout / zoneout mask generator.

The TF1 reference draws dropout and zoneout masks from TensorFlow's stateful RNG
(tf.layers.dropout / tf.nn.dropout: reference modules/self_attention.py:61,
modules/multi_speaker_modules.py:31, and the external PreNet / ZoneoutLSTMCell), which cannot
be reproduced bit-for-bit outside TF (SURVEY.md §7 "Dropout/zoneout RNG").  The build therefore
defines its own stateless mask:  keep(seed, stream, idx) = lowbias32(idx ^ seed*K1 + stream*K2) >= rate*2^32,
implemented identically here (numpy) and in csrc/common.h (HIP) so that parity tests can run
with dropout / zoneout switched ON and still compare element-wise.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

# stream ids: one per dropout / zoneout site of the hot path (must match csrc/common.h)
STREAM_ENC_PRENET0 = 1
STREAM_ENC_PRENET1 = 2
STREAM_ENC_LSTM_FW_C = 3
STREAM_ENC_LSTM_FW_H = 4
STREAM_ENC_LSTM_BW_C = 5
STREAM_ENC_LSTM_BW_H = 6
STREAM_ENC_SA = 7
STREAM_DEC_PRENET0 = 8
STREAM_DEC_PRENET1 = 9
STREAM_ATT_LSTM_C = 10
STREAM_ATT_LSTM_H = 11
STREAM_LSTM1_C = 12
STREAM_LSTM1_H = 13
STREAM_LSTM2_C = 14
STREAM_LSTM2_H = 15
STREAM_DEC_SA = 16
STREAM_POSTNET0 = 17  # +layer

_M32 = np.uint64(0xFFFFFFFF)


def hash_u32(seed, stream, idx):
    """lowbias32 of (idx ^ seed*0x9E3779B1) + stream*0x85EBCA6B, all mod 2^32. idx: integer array."""
    idx = np.asarray(idx, dtype=np.uint64) & _M32
    s = (np.uint64(seed) * np.uint64(0x9E3779B1)) & _M32
    x = (idx ^ s)
    x = (x + ((np.uint64(stream) * np.uint64(0x85EBCA6B)) & _M32)) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def rate_threshold(rate):
    """uint32 threshold t such that keep <=> hash >= t ; P(drop) = t / 2^32."""
    if rate <= 0.0:
        return np.uint32(0)
    return np.uint32(min(int(np.float32(rate) * np.float32(4294967296.0)), 0xFFFFFFFF))


def keep_mask(seed, stream, shape, rate):
    """Boolean keep-mask of `shape`; element index = C-order linear index."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    h = hash_u32(seed, stream, idx)
    return (h >= rate_threshold(rate)).reshape(shape)
