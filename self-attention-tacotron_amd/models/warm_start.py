"""Warm start from / export to TensorFlow checkpoints (reference train.py:76-78: tf.estimator.WarmStartSettings(
ckpt_to_initialize_from=hparams.ckpt_to_initialize_from, vars_to_warm_start=hparams.vars_to_warm_start); hparams.py:200-202).

The reference restores by the TF graph's own variable names.  This build keeps its parameters in ONE flat buffer with several
of the reference's variables fused side by side (K|V|Q projections, highway H|T, mel|stop projections, the two query layers),
so the correspondence is an explicit VARIABLE MAP: a JSON object  { "<tf variable name>": <target>, ... }  with
    <target> = {"param": "<name in params.layout>", "rows": [r0, r1], "cols": [c0, c1]}     rows / cols optional (whole axis)
             | {"buffer": "<BatchNorm name in Engine.bn>", "stat": "mean" | "var"}           moving statistics
             | {"ignore": true}                                                               e.g. global_step, Adam slots
WITHOUT a user-written map the default correspondence is used (r3): the variable names that the in-tree source fixes
(default_var_patterns: attention mechanisms, conv bank / projections, mel / stop projections, self-attention blocks) are
matched as name suffixes against the checkpoint, the rest by shape uniqueness (resolve_default_map).  Most of the remaining
layers live in the un-vendored `tacotron2` package and are auto-named by tf.layers, so where neither rule gives a unique
answer warm start raises and names what is open: `tools/tf_checkpoint.py list` prints the names and shapes of a checkpoint,
`... template` writes a map with every parameter of this build and empty TF names, `... suggest` fills in what the default
rules resolve for YOUR checkpoint.  Shapes agree without transposition: Dense kernels [in, out], Conv1D kernels [width, in, out],
LSTMCell kernels [in + units, 4 units] with gate order i | j | f | o.

`vars_to_warm_start` keeps tf.estimator's meaning: a regular expression (or a list of them) matched with re.match against the
TF variable names of the map; every matching variable must be in the checkpoint with the mapped shape - anything else raises.
"""
import json
import logging
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
        v = engine.bn[tgt["buffer"]][0 if tgt["stat"] == "mean" else 1]
        return v[tgt["rows"][0]:tgt["rows"][1]] if "rows" in tgt else v
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


def default_var_patterns(cfg):
    """The part of the correspondence that IS derivable from the in-tree source: [(regular expression over TF variable names,
    target)].  A pattern is a name SUFFIX - the enclosing scopes (estimator / layer-class scopes of the un-vendored tacotron2
    package) are left open - and is resolved against a concrete checkpoint by resolve_default_map(): it must select exactly
    one variable of the target's shape.  Sources of the names:
      * modules/forward_attention.py:17-24  tf.get_variable("attention_variable" / "attention_bias") inside
        variable_scope(None, "location_sensitive_attention") (:89); :73 Conv1D name="location_features_convolution";
        :78 Dense name="location_features_layer"; :86 Dense name="transition_factor_projection"; the query / memory layers
        are BahdanauAttention's "query_layer" / "memory_layer" (two mechanisms: the second instance is uniquified) and its
        score variable "attention_v" under "bahdanau_attention" (modules/attentions.py:53-57)
      * modules/module.py:51,60,67          Conv1d(name="conv1d_K<k>" | "proj1" | "proj2"): kernel + batch-norm scale / offset /
        moving statistics below that scope (one BatchNorm per bank width: this build keeps them side by side)
      * modules/module.py:717-724           Projection(name="out_projection" | "stop_token_projection") under scope "decoder":
        tf.get_variable('kernel' / 'bias') (:629-637)
      * modules/self_attention.py:102-106   four unnamed Dense layers of MultiHeadAttention, uniquified in CALL order
        (:113-126): dense = key, dense_1 = value, dense_2 = query, dense_3 = output; encoder and decoder instances differ by
        shape.  SelfAttentionTransformer's tanh Dense (modules/module.py:359) is the one Dense directly under its scope.
        With self_attention_num_hop / decoder_self_attention_num_hop > 1 the hops of a stack have equal shapes and only TF's
        creation-order suffixes tell them apart: no pattern is offered for them - the loose ones below then select several
        variables and are dropped, so such a warm start asks for a written variable map instead of guessing the order.
    Everything else (embedding, pre-nets, highway, LSTM cells, post-net: tacotron2 names) is left to shape uniqueness."""
    CC, K = cfg.conv_channels, cfg.max_filter_width
    pats = []
    lsa, bah = r"(.*/)?location_sensitive_attention(_\d+)?/", r"(.*/)?bahdanau_attention(_\d+)?/"
    pats += [(lsa + r"attention_variable$", {"param": "dec.att1.v"}), (lsa + r"attention_bias$", {"param": "dec.att1.b"}),
             (r".*location_features_convolution/kernel$", {"param": "dec.att1.F"}),
             (r".*location_features_convolution/bias$", {"param": "dec.att1.bF"}),
             (r".*location_features_layer/kernel$", {"param": "dec.att1.U"}),
             (r".*memory_layer(_\d+)?/kernel$", {"param": "dec.att1.Wm"})]
    U1, U2 = cfg.att1_units, cfg.att2_units
    pats.append((r".*(location_sensitive_attention|ForwardAttention).*query_layer(_\d+)?/kernel$",
                 {"param": "dec.att.Wq", "cols": [0, U1]}))
    if cfg.transition_agent:
        pats += [(r".*transition_factor_projection/kernel$", {"param": "dec.att1.Wa"}),
                 (r".*transition_factor_projection/bias$", {"param": "dec.att1.ba"})]
    if cfg.dual:
        pats += [(bah + r"attention_v$", {"param": "dec.att2.v"}),
                 (r".*memory_layer(_\d+)?/kernel$", {"param": "dec.att2.Wm"}),
                 (r".*(bahdanau_attention|BahdanauAttention).*query_layer(_\d+)?/kernel$",
                  {"param": "dec.att.Wq", "cols": [U1, U1 + U2]})]
    bn = (("gamma", "gamma"), ("beta", "beta"))
    for k in range(1, K + 1):
        sc = r"(.*/)?conv1d_K%d/" % k
        pats.append((sc + r".*kernel$", {"param": "enc.bank%d.W" % k}))
        for tfn, ours in bn:
            pats.append((sc + r".*%s$" % tfn, {"param": "enc.bank." + ours, "rows": [(k - 1) * CC, k * CC]}))
        pats += [(sc + r".*moving_mean$", {"buffer": "bank", "stat": "mean", "rows": [(k - 1) * CC, k * CC]}),
                 (sc + r".*moving_variance$", {"buffer": "bank", "stat": "var", "rows": [(k - 1) * CC, k * CC]})]
    for name in ("proj1", "proj2"):
        sc = r"(.*/)?%s/" % name
        pats.append((sc + r".*kernel$", {"param": "enc.%s.W" % name}))
        for tfn, ours in bn:
            pats.append((sc + r".*%s$" % tfn, {"param": "enc.%s.%s" % (name, ours)}))
        pats += [(sc + r".*moving_mean$", {"buffer": name, "stat": "mean"}),
                 (sc + r".*moving_variance$", {"buffer": name, "stat": "var"})]
    W = cfg.num_mels * cfg.r
    pats += [(r"(.*/)?decoder/out_projection/kernel$", {"param": "dec.out.W", "cols": [0, W]}),
             (r"(.*/)?decoder/out_projection/bias$", {"param": "dec.out.b", "rows": [0, W]}),
             (r"(.*/)?decoder/stop_token_projection/kernel$", {"param": "dec.out.W", "cols": [W, W + 1]}),
             (r"(.*/)?decoder/stop_token_projection/bias$", {"param": "dec.out.b", "rows": [W, W + 1]})]
    for pre, s in (("enc.sa", cfg.sa_units), ("dec.sa", cfg.dec_sa_units)):
        if not s:
            continue
        mha = r"(.*/)?multi_head_attention(_\d+)?/"
        for j, (leaf, c0) in enumerate((("dense", 0), ("dense_1", s), ("dense_2", 2 * s))):     # key | value | query
            pats += [(mha + leaf + r"/kernel$", {"param": pre + ".kvq.W", "cols": [c0, c0 + s]}),
                     (mha + leaf + r"/bias$", {"param": pre + ".kvq.b", "rows": [c0, c0 + s]})]
        pats += [(mha + r"dense_3/kernel$", {"param": pre + ".o.W"}), (mha + r"dense_3/bias$", {"param": pre + ".o.b"}),
                 (r"(.*/)?self_attention_transformer(_\d+)?/dense(_\d+)?/kernel$", {"param": pre + ".t.W"}),
                 (r"(.*/)?self_attention_transformer(_\d+)?/dense(_\d+)?/bias$", {"param": pre + ".t.b"})]
    return pats


def _target_shape(engine, tgt):
    return tuple(_target_view(engine, "?", tgt).shape)


def _is_slot(name):
    """optimizer slots and counters of a training checkpoint"""
    return name == "global_step" or name.endswith(("/Adam", "/Adam_1")) or name.split("/")[-1] in ("beta1_power", "beta2_power")


class ShapeEngine:
    """stand-in for engine.Engine where only the SHAPES of parameters and statistics matter (CPU tools)"""

    def __init__(self, cfg):
        from ..params import param_shapes
        self.cfg = cfg
        self.P = {k: torch.empty(s) for k, s in param_shapes(cfg)}
        nb = cfg.max_filter_width * cfg.conv_channels
        self.bn = {n: (torch.empty(c), torch.empty(c)) for n, c in (("bank", nb), ("proj1", cfg.proj1), ("proj2", cfg.proj2))}
        if cfg.use_postnet_v2:
            for n in range(cfg.num_postnet_v2_layers):
                self.bn["postnet%d" % n] = (torch.empty(cfg.postnet_v2_out_channels), torch.empty(cfg.postnet_v2_out_channels))


def all_targets(engine):
    """every atomic target of this configuration in canonical form: whole parameters, the slices of the fused ones
    (fused_slices; 1-D tensors are sliced by "rows"), one slice per bank width of the conv bank's BatchNorm, the moving statistics"""
    from ..params import param_shapes
    cfg = engine.cfg
    CC, K = cfg.conv_channels, cfg.max_filter_width
    fs = fused_slices(cfg)
    out = []
    for name, shp in param_shapes(cfg):
        if name in fs:
            for what, rows, cols in fs[name]:
                t = {"param": name}
                if rows:
                    t["rows"] = list(rows)
                if cols:
                    t["rows" if len(shp) == 1 else "cols"] = list(cols)
                out.append(t)
        elif name in ("enc.bank.gamma", "enc.bank.beta"):
            out += [{"param": name, "rows": [(k - 1) * CC, k * CC]} for k in range(1, K + 1)]
        else:
            out.append({"param": name})
    for b in engine.bn:
        for st in ("mean", "var"):
            if b == "bank":
                out += [{"buffer": b, "stat": st, "rows": [(k - 1) * CC, k * CC]} for k in range(1, K + 1)]
            else:
                out.append({"buffer": b, "stat": st})
    return out


def resolve_default_map(engine, reader):
    """A concrete variable map for THIS checkpoint without a user-written one.  (1) every pattern of default_var_patterns
    that selects exactly one not-yet-used checkpoint variable of its target's shape (singleton axes ignored) is taken; (2) of
    the targets still open, those whose shape occurs exactly once among the remaining checkpoint variables AND once among the
    remaining targets are paired (shape uniqueness).  Returns (var_map, unresolved targets)."""
    squeeze = lambda shp: tuple(d for d in shp if d != 1) or (1,)
    key = lambda t: json.dumps(t, sort_keys=True)
    ck = {n: squeeze(tuple(reader.entries[n]["shape"])) for n in reader.entries if not _is_slot(n)}
    targets = {key(t): t for t in all_targets(engine)}
    used, vmap, done = set(), {}, set()
    for pat, tgt in default_var_patterns(engine.cfg):
        k = key(tgt)
        if k in done or k not in targets:
            continue
        want = squeeze(_target_shape(engine, tgt))
        hits = [n for n in ck if n not in used and ck[n] == want and re.match(pat, n)]
        if len(hits) == 1:
            vmap[hits[0]] = tgt; used.add(hits[0]); done.add(k)
    open_t = [t for k, t in targets.items() if k not in done]
    shapes_t = [squeeze(_target_shape(engine, t)) for t in open_t]
    unresolved = []
    for t, want in zip(open_t, shapes_t):
        hits = [n for n in ck if n not in used and ck[n] == want]
        # shape uniqueness alone can bind a variable of a layer this build does not model: the LEAF of the TF name must also
        # agree with the kind of the target (kernel / bias / gamma / beta / moving statistics / embedding table), and every
        # such pairing is logged with both names so a wrong initialisation cannot pass silently
        if len(hits) == 1 and shapes_t.count(want) == 1 and _leaf_agrees(hits[0], t, want):
            vmap[hits[0]] = t; used.add(hits[0])
            logging.warning("warm start: %s <- TF variable %r paired by SHAPE %s only (no name pattern matched); check it, or "
                            "pin it in a variable map (tools/tf_checkpoint.py suggest)", _target_name(t), hits[0], list(want))
        else:
            unresolved.append(t)
    return vmap, unresolved


def _target_name(t):
    return t.get("param") or ("%s/%s" % (t["buffer"], t["stat"]))


def _leaf_agrees(tf_name, tgt, shape):
    """minimal name sanity of a shape-only pairing: the last path component of the TF variable against the target's kind.
    Leaves TF1 layers are known to use are classified; an unclassified leaf (e.g. attention_v) may only bind a plain
    matrix / vector target of the same rank."""
    leaf = tf_name.rsplit("/", 1)[-1].lower()
    known = {"kernel": "matrix", "weights": "matrix", "bias": "vector", "biases": "vector", "gamma": "gamma", "beta": "beta",
             "moving_mean": "mean", "moving_variance": "var"}
    tk = known.get(leaf) or ("embedding" if "embedding" in tf_name.lower() else None)
    if "buffer" in tgt:
        want = "mean" if tgt["stat"] == "mean" else "var"
    else:
        last = tgt["param"].rsplit(".", 1)[-1]
        want = last if last in ("gamma", "beta") else "embedding" if tgt["param"].endswith("embedding") else \
            "matrix" if len(shape) >= 2 else "vector"
    if tk is None:
        return want in ("matrix", "vector")
    return tk == want


def warm_start(engine, ckpt_to_initialize_from, vars_to_warm_start, var_map=None):
    """copy the selected variables of a TensorFlow checkpoint into the engine's parameters / BatchNorm statistics.
    Returns the list of TF variable names loaded.  Without a user-written variable map the default correspondence is
    resolved against the checkpoint (resolve_default_map); with the default `vars_to_warm_start=[".*"]` EVERY parameter must
    then be resolved - tf.estimator raises for a model variable that is not in the checkpoint, and so does this."""
    reader = CheckpointReader(ckpt_to_initialize_from)
    if not var_map:
        var_map, unresolved = resolve_default_map(engine, reader)
        pats = [vars_to_warm_start] if isinstance(vars_to_warm_start, str) else list(vars_to_warm_start or [".*"])
        if unresolved and any(p in (".*", "") for p in pats):
            what = ", ".join(sorted({t.get("param") or ("%s/%s" % (t["buffer"], t["stat"])) for t in unresolved}))
            raise UnsupportedConfiguration(
                "warm start: no unique TensorFlow variable in %s for: %s.  These names live in the un-vendored tacotron2 "
                "package: write them into a variable map (tools/tf_checkpoint.py suggest <ckpt> > map.json, hparam "
                "warm_start_var_map=map.json) or narrow vars_to_warm_start" % (ckpt_to_initialize_from, what))
    names = matching(var_map, vars_to_warm_start)
    if not names:
        raise ValueError("vars_to_warm_start=%r selects no variable of the map" % (vars_to_warm_start,))
    for n in names:
        if not reader.has_tensor(n):
            raise CheckpointError("warm start: variable %r is not in the checkpoint %s" % (n, ckpt_to_initialize_from))
        a = reader.get_tensor(n)
        dst = _target_view(engine, n, var_map[n])
        if tuple(a.shape) != tuple(dst.shape):
            if int(np.prod(a.shape)) == int(np.prod(dst.shape)) and \
                    [d for d in a.shape if d != 1] == [d for d in dst.shape if d != 1]:
                a = np.asarray(a).reshape(tuple(dst.shape))            # singleton axes only (e.g. [out] vs [out, 1])
            else:
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
    from ..params import sa_prefixes
    for pre, s, _ in sa_prefixes(cfg):      # every hop of both stacks
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
