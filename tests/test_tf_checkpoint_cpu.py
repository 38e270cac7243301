"""CPU tests of the TensorFlow-checkpoint reader / writer and of the warm-start variable map (reference
train.py:76-78, hparams.py:200-202).  No TensorFlow exists here, so the files are checked against the published format's
known answers (LevelDB table magic, CRC-32C check value, masked-CRC definition) and by round trips through the writer."""
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest
import torch

import satt_amd  # noqa: F401
from satt_amd.models import warm_start as ws
from satt_amd.params import ModelConfig, init_params, param_shapes
from satt_amd.utils import tf_checkpoint as tc
from satt_amd.utils import tfrecord

from common import SMALL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_crc32c_known_answers_and_block_combination():
    assert tc.crc32c(b"123456789") == 0xE3069283            # the CRC-32C check value (RFC 3720 appendix B.4 family)
    assert tc.crc32c(bytes(32)) == 0x8A9136AA               # RFC 3720 B.4: 32 bytes of zeros
    assert tc.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43      # RFC 3720 B.4: 32 bytes of ones
    g = np.random.default_rng(0)
    for n in (1, 1000, 16384, 16385, 70001, 300000):        # the blocked numpy path against the byte-serial one
        b = g.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert tc.crc32c(b) == tfrecord.crc32c(b), n
    assert tc.masked_crc(b"abc") == tfrecord.masked_crc(b"abc")


def test_bundle_round_trip_multi_block_and_layout(tmp_path):
    g = np.random.default_rng(1)
    T = {"model/dense/kernel": g.normal(size=(7, 5)).astype(np.float32), "model/dense/bias": np.arange(5, dtype=np.float32),
         "global_step": np.array(123, dtype=np.int64), "flags": np.array([True, False]),
         **{"scope_%03d/w" % i: g.normal(size=(2, i % 7 + 1)).astype(np.float32) for i in range(400)}}
    prefix = str(tmp_path / "model.ckpt-123")
    tc.write_checkpoint(prefix, T)
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) > 2 * 4096      # several data blocks
    r = tc.CheckpointReader(prefix)
    assert r.get_variable_to_shape_map()["model/dense/kernel"] == [7, 5] and r.get_variable_to_shape_map()["global_step"] == []
    for k, v in T.items():
        got = r.get_tensor(k)
        assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v), k
    keys = [k for k, _ in tc.read_table(prefix + ".index")]
    assert keys[0] == b"" and keys == sorted(keys)
    with pytest.raises(tc.CheckpointError, match="not in the checkpoint"):
        r.get_tensor("nope")


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "m")
    tc.write_checkpoint(prefix, {"a": np.arange(100, dtype=np.float32), "b": np.ones((3, 3), np.float32)})
    d = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); d[17] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(d))
    with pytest.raises(tc.CheckpointError, match="checksum"):
        tc.CheckpointReader(prefix).get_tensor("a")
    assert np.array_equal(tc.CheckpointReader(prefix).get_tensor("b"), np.ones((3, 3), np.float32))
    i = bytearray(open(prefix + ".index", "rb").read()); i[5] ^= 1
    open(prefix + ".index", "wb").write(bytes(i))
    with pytest.raises(tc.CheckpointError, match="checksum"):
        tc.CheckpointReader(prefix)
    open(prefix + ".index", "wb").write(b"x" * 100)
    with pytest.raises(tc.CheckpointError, match="magic"):
        tc.CheckpointReader(prefix)


class FakeEngine:
    """the part of Engine the warm start touches (views of parameters and BatchNorm statistics), on the CPU"""

    def __init__(self, cfg, seed):
        self.cfg = cfg
        self.P = {k: torch.as_tensor(v).clone() for k, v in init_params(cfg, seed).items()}
        nb = cfg.max_filter_width * cfg.conv_channels
        g = torch.Generator().manual_seed(seed)
        self.bn = {n: (torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5)
                   for n, c in (("bank", nb), ("proj1", cfg.proj1), ("proj2", cfg.proj2))}
        self.refreshed = 0

    def refresh_shadows(self):
        self.refreshed += 1


def test_warm_start_through_a_variable_map(tmp_path):
    cfg = ModelConfig(**SMALL)
    vmap = {k.replace("?/", "tf/"): v for k, v in ws.template(cfg).items() if k != "_comment"}
    # the map covers every parameter exactly once (fused tensors through their column slices)
    cover = {n: np.zeros(s, dtype=np.int32) for n, s in param_shapes(cfg)}
    for t in vmap.values():
        if "param" in t:
            v = cover[t["param"]]
            if "cols" in t:
                v = v[..., t["cols"][0]:t["cols"][1]]
            v += 1
    assert all((c == 1).all() for c in cover.values())
    src, dst = FakeEngine(cfg, 1), FakeEngine(cfg, 2)
    prefix = str(tmp_path / "model.ckpt-7")
    names = ws.export_tf_checkpoint(src, prefix, vmap, global_step=7)
    r = tc.CheckpointReader(prefix)
    assert int(r.get_tensor("global_step")) == 7 and "tf/dec.sa.kvq.W/key_projection/kernel" in names
    S2 = cfg.dec_sa_units
    assert r.get_variable_to_shape_map()["tf/dec.sa.kvq.W/value_projection/kernel"] == [S2, S2]
    # (1) a regular expression selects the encoder only (tf.estimator vars_to_warm_start semantics: re.match on the TF names)
    got = ws.warm_start(dst, prefix, ["tf/enc\\.", "tf/embedding"], vmap)
    assert got and all(n.startswith(("tf/enc.", "tf/embedding")) for n in got) and dst.refreshed == 1
    assert torch.equal(dst.P["enc.sa.kvq.W"], src.P["enc.sa.kvq.W"]) and torch.equal(dst.P["embedding"], src.P["embedding"])
    assert not torch.equal(dst.P["dec.lstm1.W"], src.P["dec.lstm1.W"])
    assert not torch.equal(dst.bn["bank"][0], src.bn["bank"][0])           # "tf/bank/moving_mean" does not match the patterns
    # (2) everything
    ws.warm_start(dst, prefix, ".*", vmap)
    assert all(torch.equal(dst.P[k], src.P[k]) for k in src.P)
    assert all(torch.equal(dst.bn[k][i], src.bn[k][i]) for k in src.bn for i in (0, 1))
    # (3) loud failures: missing variable, shape mismatch, no map, nothing selected
    bad = dict(vmap); bad["tf/not_there"] = {"param": "embedding"}
    with pytest.raises(tc.CheckpointError, match="not in the checkpoint"):
        ws.warm_start(dst, prefix, ".*", bad)
    bad = dict(vmap); bad["tf/embedding"] = {"param": "dec.out.b"}
    with pytest.raises(tc.CheckpointError, match="shape"):
        ws.warm_start(dst, prefix, "tf/embedding", bad)
    with pytest.raises(ValueError, match="no unique TensorFlow variable"):      # no map: the default rules cannot resolve "tf/<our name>"
        ws.warm_start(dst, prefix, ".*", None)
    with pytest.raises(ValueError, match="selects no variable"):
        ws.warm_start(dst, prefix, "zzz", vmap)


def test_checkpoint_tool_list_and_suggest(tmp_path):
    cfg = ModelConfig()
    P = init_params(cfg, 0)
    # a "TF" checkpoint holding three uniquely shaped variables of the LJSpeech model under foreign names + Adam slots
    T = {"model/embedding/embedding": P["embedding"], "model/attention_lstm/kernel": P["dec.att_lstm.W"],
         "model/decoder/lstm1/kernel": P["dec.lstm1.W"],
         "model/decoder/lstm1/kernel/Adam": P["dec.lstm1.W"], "global_step": np.array(5, dtype=np.int64),
         "model/enc/proj1/kernel": P["enc.proj1.W"]}
    prefix = str(tmp_path / "model.ckpt-5")
    tc.write_checkpoint(prefix, T)
    cfgf = os.path.join(ROOT, "examples", "ljspeech", "self-attention-tacotron.json")
    tool = os.path.join(ROOT, "tools", "tf_checkpoint.py")
    out = subprocess.run([sys.executable, tool, "list", prefix], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "model/decoder/lstm1/kernel" in out.stdout and "[800, 1024]" in out.stdout
    d = json.load(open(cfgf)); d.pop("_comment", None)
    clean = str(tmp_path / "cfg.json"); json.dump(d, open(clean, "w"))
    out = subprocess.run([sys.executable, tool, "suggest", prefix, "--hparam-json-file", clean], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    m = json.loads(out.stdout)
    assert m["model/attention_lstm/kernel"] == {"param": "dec.att_lstm.W"} and m["model/decoder/lstm1/kernel"] == {"param": "dec.lstm1.W"}
    assert m["model/enc/proj1/kernel"] == {"param": "enc.proj1.W"} and "?/dec.lstm2.W" in m
    # [256, 256] occurs several times in the model (embedding, enc.prenet0.W, ...): ambiguous shapes are left to the user
    assert "model/embedding/embedding" not in m and "?/embedding" in m
    assert m["global_step"] == {"ignore": True}


def _tf_style_checkpoint(cfg, src):
    """a checkpoint under names of the form the reference's graph produces: the scopes the in-tree source fixes (attention
    mechanisms, conv bank, projections, self-attention blocks) + invented tacotron2-style scopes for the rest"""
    P, CC = src.P, cfg.conv_channels
    n = lambda a: a.detach().numpy().copy()
    S, S2, U1, U2, W = cfg.sa_units, cfg.dec_sa_units, cfg.att1_units, cfg.att2_units, cfg.num_mels * cfg.r
    dec = "model/dual_source_transformer_decoder/decoder/dual_source_attention_rnn/"
    T = {dec + "location_sensitive_attention/attention_variable": n(P["dec.att1.v"]),
         dec + "location_sensitive_attention/attention_bias": n(P["dec.att1.b"]),
         dec + "location_sensitive_attention/query_layer/kernel": n(P["dec.att.Wq"][:, :U1]),
         dec + "bahdanau_attention/query_layer/kernel": n(P["dec.att.Wq"][:, U1:]),
         dec + "bahdanau_attention/attention_v": n(P["dec.att2.v"]),
         "model/memory_layer/kernel": n(P["dec.att1.Wm"]), "model/memory_layer_1/kernel": n(P["dec.att2.Wm"]),
         dec + "location_features_convolution/kernel": n(P["dec.att1.F"]),
         dec + "location_features_convolution/bias": n(P["dec.att1.bF"]),
         dec + "location_features_layer/kernel": n(P["dec.att1.U"]),
         "model/decoder/out_projection/kernel": n(P["dec.out.W"][:, :W]), "model/decoder/out_projection/bias": n(P["dec.out.b"][:W]),
         "model/decoder/stop_token_projection/kernel": n(P["dec.out.W"][:, W:]),
         "model/decoder/stop_token_projection/bias": n(P["dec.out.b"][W:]),
         "global_step": np.array(11, dtype=np.int64)}
    enc = "model/self_attention_cbhg_encoder/zoneout_cbhg/"
    for k in range(1, cfg.max_filter_width + 1):
        sl = slice((k - 1) * CC, k * CC)
        T[enc + "conv1d_K%d/conv1d/kernel" % k] = n(P["enc.bank%d.W" % k])
        T[enc + "conv1d_K%d/batch_normalization/gamma" % k] = n(P["enc.bank.gamma"][sl])
        T[enc + "conv1d_K%d/batch_normalization/beta" % k] = n(P["enc.bank.beta"][sl])
        T[enc + "conv1d_K%d/batch_normalization/moving_mean" % k] = n(src.bn["bank"][0][sl])
        T[enc + "conv1d_K%d/batch_normalization/moving_variance" % k] = n(src.bn["bank"][1][sl])
        T[enc + "conv1d_K%d/conv1d/kernel/Adam" % k] = n(P["enc.bank%d.W" % k])          # optimizer slot: ignored
    for pj in ("proj1", "proj2"):
        T[enc + pj + "/conv1d/kernel"] = n(P["enc.%s.W" % pj])
        T[enc + pj + "/batch_normalization/gamma"] = n(P["enc.%s.gamma" % pj])
        T[enc + pj + "/batch_normalization/beta"] = n(P["enc.%s.beta" % pj])
        T[enc + pj + "/batch_normalization/moving_mean"] = n(src.bn[pj][0])
        T[enc + pj + "/batch_normalization/moving_variance"] = n(src.bn[pj][1])
    for pre, s, scope in (("enc.sa", S, "model/self_attention_cbhg_encoder/self_attention_transformer/"),
                          ("dec.sa", S2, "model/dual_source_transformer_decoder/self_attention_transformer/")):
        for j, leaf in enumerate(("dense", "dense_1", "dense_2")):
            T[scope + "self_attention/multi_head_attention/%s/kernel" % leaf] = n(P[pre + ".kvq.W"][:, j * s:(j + 1) * s])
            T[scope + "self_attention/multi_head_attention/%s/bias" % leaf] = n(P[pre + ".kvq.b"][j * s:(j + 1) * s])
        T[scope + "self_attention/multi_head_attention/dense_3/kernel"] = n(P[pre + ".o.W"])
        T[scope + "self_attention/multi_head_attention/dense_3/bias"] = n(P[pre + ".o.b"])
        T[scope + "dense/kernel"] = n(P[pre + ".t.W"])
        T[scope + "dense/bias"] = n(P[pre + ".t.b"])
    # tacotron2-side layers under invented names: resolvable only where the shape is unique
    T["model/embedding/t2_embedding"] = n(P["embedding"])
    T["model/decoder_rnn/lstm1/kernel"] = n(P["dec.lstm1.W"])
    T["model/attention_rnn/kernel"] = n(P["dec.att_lstm.W"])
    return T


def test_warm_start_with_the_default_variable_map(tmp_path):
    """f4 without a user-written map: names the in-tree reference source fixes (modules/forward_attention.py:17-24,73,78,
    modules/module.py:51,60,67,717-724, modules/self_attention.py:102-106) resolve by suffix, uniquely shaped tacotron2-side
    variables by shape, everything else is reported"""
    cfg = ModelConfig(**dict(SMALL, att1_units=12, att2_units=6, sa_units=8, dec_sa_units=20, cbhg_out_units=16, att_rnn_units=24,
                             dec_units=28, embedding_dim=18, num_symbols=23))
    src, dst = FakeEngine(cfg, 1), FakeEngine(cfg, 2)
    prefix = str(tmp_path / "model.ckpt-11")
    tc.write_checkpoint(prefix, _tf_style_checkpoint(cfg, src))
    vmap, unresolved = ws.resolve_default_map(dst, tc.CheckpointReader(prefix))
    open_params = {t.get("param") for t in unresolved}
    derivable = ["dec.att1.v", "dec.att1.b", "dec.att1.F", "dec.att1.bF", "dec.att1.U", "dec.att1.Wm", "dec.att2.Wm", "dec.att2.v",
                 "dec.att.Wq", "dec.out.W", "dec.out.b", "enc.bank.gamma", "enc.bank.beta", "enc.proj1.W", "enc.proj2.W",
                 "enc.proj1.gamma", "enc.proj2.beta", "enc.sa.kvq.W", "enc.sa.kvq.b", "enc.sa.o.W", "enc.sa.t.W", "dec.sa.kvq.W",
                 "dec.sa.o.b", "dec.sa.t.b"] + ["enc.bank%d.W" % k for k in range(1, cfg.max_filter_width + 1)]
    assert not (set(derivable) & open_params), set(derivable) & open_params
    assert not any("buffer" in t for t in unresolved)                       # every moving statistic resolved
    assert {"embedding", "dec.lstm1.W", "dec.att_lstm.W"} & open_params == set()          # unique shapes
    assert {"enc.prenet0.W", "dec.lstm2.W", "enc.highway0.W"} <= open_params   # not in the checkpoint / tacotron2 names: open
    assert not any(n.endswith("/Adam") for n in vmap)
    # ".*" must resolve EVERYTHING (tf.estimator raises for a model variable missing from the checkpoint)
    with pytest.raises(ValueError, match="enc.prenet0.W"):
        ws.warm_start(dst, prefix, [".*"], None)
    # a narrower selection loads what it names
    got = ws.warm_start(dst, prefix, ["model/.*attention.*", "model/memory_layer", "model/decoder/",
                                       "model/self_attention_cbhg_encoder/zoneout_cbhg/"], None)
    assert len(got) > 40
    for k in derivable:
        assert torch.equal(dst.P[k], src.P[k]), k
    assert all(torch.equal(dst.bn[k][i], src.bn[k][i]) for k in src.bn for i in (0, 1))
    assert not torch.equal(dst.P["embedding"], src.P["embedding"])                       # not selected
