"""Warm start from / export to TensorFlow checkpoints (reference train.py:76-78: tf.estimator.WarmStartSettings(
ckpt_to_initialize_from=hparams.ckpt_to_initialize_from, vars_to_warm_start=hparams.vars_to_warm_start); hparams.py:200-202).

The reference restores by the TF graph's own variable names.  This build keeps its parameters in ONE flat buffer with several
of the reference's variables fused side by side (K|V|Q projections, highway H|T, mel|stop projections, the two query layers),
so the correspondence is an explicit VARIABLE MAP: a JSON object  { "<tf variable name>": <target>, ... }  with
    <target> = {"param": "<name in params.layout>", "rows": [r0, r1], "cols": [c0, c1]}     rows / cols optional (whole axis)
             | {"buffer": "<BatchNorm name in Engine.bn>", "stat": "mean" | "var"}           moving statistics
             | {"ignore": true}                                                               e.g. global_step, Adam slots
Most of the reference's layers live in the un-vendored `tacotron2` package and are auto-named by tf.layers, so their variable
names cannot be derived from /root/reference alone: `tools/tf_checkpoint.py list` prints the names and shapes of a
checkpoint, `... template` writes a map with every parameter of this build and empty TF names, `... suggest` fills in the
names whose shape is unique.  Shapes agree without transposition: Dense kernels [in, out], Conv1D kernels [width, in, out],
LSTMCell kernels [in + units, 4 units] with gate order i | j | f | o.

`vars_to_warm_start` keeps tf.estimator's meaning: a regular expression (or a list of them) matched with re.match against the
TF variable names of the map; every matching variable must be in the checkpoint with the mapped shape - anything else raises.
"""
import json
import re

import numpy as np
import torch

from ..modules.attentions import UnsupportedConfiguration
from ..utils.tf_checkpoint import CheckpointError, CheckpointReader, write_checkpoint


def load_var_map(path):
    m = json.load(open(path))
    m.pop("_comment", None)
    for name, tgt in m.items():
        if not isinstance(tgt, dict) or not (("param" in tgt) or ("buffer" in tgt) or tgt.get("ignore")):
            raise ValueError("variable map entry %r needs one of 'param', 'buffer', 'ignore'" % name)
    return m


def _target_view(engine, name, tgt):
    """the tensor view of the engine a map entry points at"""
    if "buffer" in tgt:
        if tgt["buffer"] not in engine.bn or tgt.get("stat") not in ("mean", "var"):
            raise ValueError("variable map entry %r: unknown BatchNorm buffer %r / stat %r" % (name, tgt["buffer"], tgt.get("stat")))
        return engine.bn[tgt["buffer"]][0 if tgt["stat"] == "mean" else 1]
    if tgt["param"] not in engine.P:
        raise ValueError("variable map entry %r: no parameter %r in this configuration" % (name, tgt["param"]))
    v = engine.P[tgt["param"]]
    if "rows" in tgt:
        v = v[tgt["rows"][0]:tgt["rows"][1]]
    if "cols" in tgt:
        v = v[..., tgt["cols"][0]:tgt["cols"][1]]
    return v


def matching(var_map, vars_to_warm_start):
    pats = [vars_to_warm_start] if isinstance(vars_to_warm_start, str) else list(vars_to_warm_start or [".*"])
    return [n for n, t in var_map.items() if not t.get("ignore") and any(re.match(p, n) for p in pats)]


def warm_start(engine, ckpt_to_initialize_from, vars_to_warm_start, var_map):
    """copy the selected variables of a TensorFlow checkpoint into the engine's parameters / BatchNorm statistics.
    Returns the list of TF variable names loaded."""
    if not var_map:
        raise UnsupportedConfiguration("warm_start needs a variable map (hparam warm_start_var_map=<json>): the reference's TF "
                                       "variable names cannot be derived without TensorFlow - see models/warm_start.py")
    reader = CheckpointReader(ckpt_to_initialize_from)
    names = matching(var_map, vars_to_warm_start)
    if not names:
        raise ValueError("vars_to_warm_start=%r selects no variable of the map" % (vars_to_warm_start,))
    for n in names:
        if not reader.has_tensor(n):
            raise CheckpointError("warm start: variable %r is not in the checkpoint %s" % (n, ckpt_to_initialize_from))
        a = reader.get_tensor(n)
        dst = _target_view(engine, n, var_map[n])
        if tuple(a.shape) != tuple(dst.shape):
            raise CheckpointError("warm start: %r has shape %s in the checkpoint, the mapped target %s has %s"
                                  % (n, list(a.shape), var_map[n], list(dst.shape)))
        dst.copy_(torch.as_tensor(np.asarray(a, dtype=np.float32)))
    engine.refresh_shadows()
    return names


def export_tf_checkpoint(engine, prefix, var_map, global_step=0):
    """the inverse: write the engine's parameters (and moving statistics) under the TF names of the map, plus global_step"""
    out = {"global_step": np.array(int(global_step), dtype=np.int64)}
    for n, tgt in var_map.items():
        if tgt.get("ignore"):
            continue
        out[n] = _target_view(engine, n, tgt).detach().float().cpu().numpy()
    write_checkpoint(prefix, out)
    return sorted(out)


def fused_slices(cfg):
    """the parameters of this build that hold SEVERAL reference variables side by side: param -> [(what, rows, cols)];
    everything not listed here maps one to one (reference lines: where the separate layers are created)."""
    S, S2, H = cfg.sa_units, cfg.dec_sa_units, cfg.cbhg_out_units // 2
    out = {}
    for pre, s in (("enc.sa", S), ("dec.sa", S2)):
        if s:       # modules/self_attention.py:103-106: key / value / query projections are three Dense layers
            out[pre + ".kvq.W"] = [("key_projection/kernel", None, (0, s)), ("value_projection/kernel", None, (s, 2 * s)),
                                   ("query_projection/kernel", None, (2 * s, 3 * s))]
            out[pre + ".kvq.b"] = [("key_projection/bias", None, (0, s)), ("value_projection/bias", None, (s, 2 * s)),
                                   ("query_projection/bias", None, (2 * s, 3 * s))]
    for n in range(cfg.num_highway):      # tacotron2 HighwayNet: H and T Dense layers
        out[f"enc.highway{n}.W"] = [("H/kernel", None, (0, H)), ("T/kernel", None, (H, 2 * H))]
        out[f"enc.highway{n}.b"] = [("H/bias", None, (0, H)), ("T/bias", None, (H, 2 * H))]
    U1, U2 = cfg.att1_units, cfg.att2_units
    if U2:      # one query layer per mechanism (modules/forward_attention.py:92, BahdanauAttention query_layer)
        out["dec.att.Wq"] = [("ForwardAttention/query_layer/kernel", None, (0, U1)),
                             ("BahdanauAttention/query_layer/kernel", None, (U1, U1 + U2))]
    W = cfg.num_mels * cfg.r    # modules/module.py:717-723: out_projection and stop_token_projection
    out["dec.out.W"] = [("out_projection/kernel", None, (0, W)), ("stop_token_projection/kernel", None, (W, W + 1))]
    out["dec.out.b"] = [("out_projection/bias", None, (0, W)), ("stop_token_projection/bias", None, (W, W + 1))]
    return out


def template(cfg):
    """a variable map with every parameter / statistic of this configuration and placeholder TF names ("?/..."): fill in
    the names of YOUR checkpoint (tools/tf_checkpoint.py list / suggest)"""
    from ..params import param_shapes
    fs = fused_slices(cfg)
    m = {"_comment": "replace every '?/...' key by the TensorFlow variable name of your checkpoint; see models/warm_start.py",
         "global_step": {"ignore": True}}
    for name, shp in param_shapes(cfg):
        if name in fs:
            for what, rows, cols in fs[name]:
                t = {"param": name}
                if rows:
                    t["rows"] = list(rows)
                if cols:
                    t["cols"] = list(cols)
                m["?/%s/%s" % (name, what)] = t
        else:
            m["?/%s" % name] = {"param": name}
    nb = ["bank", "proj1", "proj2"] + ([f"postnet{n}" for n in range(cfg.num_postnet_v2_layers)] if cfg.use_postnet_v2 else [])
    for b in nb:
        m["?/%s/moving_mean" % b] = {"buffer": b, "stat": "mean"}
        m["?/%s/moving_variance" % b] = {"buffer": b, "stat": "var"}
    return m
