"""Parameter layout of the MI355X build: ONE flat fp32 buffer (6 246 104 floats for the LJSpeech config) with
named views, so that the optimiser is a single fused kernel and the data-parallel exchange is a few large
RCCL all-reduces over contiguous slices.  Weight layouts follow SURVEY.md Appendix A: Dense W:[in,out];
Conv1D kernel [k,in,out]; LSTM kernel [in+h,4h] with gate column blocks i,j,f,o; the K/V/Q (and H/T, and Wq1/Wq2,
and mel/stop) projections are stored fused ([in, K|V|Q] ...) — `to_reference_names` documents the mapping back to
the reference's variables (checkpoint interop is a "next" row)."""
import math

import numpy as np


class ModelConfig:
    """Resolved model dimensions (from hparams; defaults = examples/ljspeech/self-attention-tacotron.json)."""

    def __init__(self, **kw):
        self.num_symbols = 256; self.embedding_dim = 256
        self.enc_prenet = (256, 128); self.enc_prenet_drop = 0.5
        self.conv_channels = 128; self.max_filter_width = 16
        self.proj1 = 128; self.proj2 = 128; self.num_highway = 4; self.cbhg_out_units = 256
        self.sa_units = 32; self.sa_heads = 2; self.sa_drop = 0.05
        # self_attention_num_hop / decoder_self_attention_num_hop (modules/module.py:411-419, :707-715): stacked
        # SelfAttentionTransformer blocks, each with its own weights; parameters of hop h: sa_prefix(base, h)
        self.sa_num_hop = 1; self.dec_sa_num_hop = 1
        self.dec_prenet = (256, 128); self.dec_prenet_drop = 0.5
        # apply_dropout_on_inference (reference hparams.py, modules/module.py:564-577): the plain decoder PreNet layers keep their
        # dropout in evaluation and synthesis
        self.apply_dropout_on_inference = False
        self.att_rnn_units = 256; self.att1_units = 224; self.att2_units = 32
        self.att_kernel = 10; self.att_filters = 5
        # first-source mechanism (modules/attentions.py:25-62): "forward" (alpha recursion, modules/forward_attention.py
        # :104-110) or "location_sensitive" (same score, plain softmax alignments); cumulative_weights feeds the location
        # convolution with the running SUM of the softmax alignments instead of the last one (:118-121)
        self.attention = "forward"; self.cumulative_weights = False
        # use_forward_attention_transition_agent (modules/forward_attention.py:80-86,111-116): u of the forward recursion is
        # predicted per step by Dense(1, sigmoid)([context | processed query]) instead of staying 0.5
        self.transition_agent = False
        # use_l2_regularization / l2_regularization_weight (reference hparams.py:175-176): read by the baseline model_fn only
        # (models/models.py:109-114); 0 = off
        self.l2_weight = 0.0
        self.dec_units = 256; self.dec_sa_units = 256; self.dec_sa_heads = 2; self.dec_sa_drop = 0.05
        self.num_mels = 80; self.r = 2; self.n_feed_frame = 2
        self.zc = 0.1; self.zh = 0.1
        self.bn_eps = 1e-3; self.bn_momentum = 0.99
        self.num_speakers = 0; self.speaker_dim = 16; self.speaker_offset = 0
        # optional PostNetV2 (reference hparams.py:158-162, models/models.py:92-100; off in the shipped configs)
        self.use_postnet_v2 = False; self.num_postnet_v2_layers = 5; self.postnet_v2_kernel_size = 5
        self.postnet_v2_out_channels = 512; self.postnet_v2_drop_rate = 0.5
        for k, v in kw.items():
            if not hasattr(self, k):
                raise KeyError(k)
            setattr(self, k, v)
        if self.sa_num_hop < 1 or self.dec_sa_num_hop < 1:
            raise ValueError("sa_num_hop / dec_sa_num_hop must be >= 1")
        if (self.sa_units > 0) != (self.att2_units > 0):
            raise ValueError("sa_units and att2_units are both zero (single attention source) or both positive")

    @property
    def ctx_dim(self):
        return self.cbhg_out_units + self.sa_units

    # sa_units = att2_units = 0: ONE attention source (ZoneoutEncoderV1 + AttentionRNN of ExtendedDecoder, reference
    # modules/module.py:293-342,530-623); dec_sa_units = 0: no decoder self-attention block, mel / stop projections read
    # the DecoderRNNV2 output (OutputAndStopTokenWrapper).  All three zero = the baseline ExtendedTacotronV1Model.
    @property
    def dual(self):
        return self.sa_units > 0

    @property
    def out_in(self):
        return self.dec_sa_units if self.dec_sa_units > 0 else self.dec_units

    @classmethod
    def from_hparams(cls, hp):
        """Map the reference's hparams (hparams.py:10-226) onto model dimensions
        (reference models/models.py:1200-1217 encoder_factory, :1318-1340 decoder_factory).  Model / encoder / decoder /
        attention strings go through the reference-named factories first: unknown names raise ValueError exactly as the
        reference does, known-but-unbuilt ones raise UnsupportedConfiguration - nothing is silently substituted."""
        from .models.models import validate_params
        validate_params(hp)
        baseline = hp.tacotron_model == "ExtendedTacotronV1Model"      # models/models.py:20-226, attention_factories.py:11-20
        return cls(**(dict(sa_units=0, att2_units=0, dec_sa_units=0, att1_units=hp.attention_out_units) if baseline else
                      dict(sa_units=hp.self_attention_out_units, att2_units=hp.attention2_out_units,
                           dec_sa_units=hp.decoder_self_attention_out_units, att1_units=hp.attention1_out_units)),
            attention=hp.attention, cumulative_weights=bool(hp.cumulative_weights),
            transition_agent=bool(hp.use_forward_attention_transition_agent) and hp.attention == "forward",
            # the dual-source model_fn never reads use_l2_regularization (models/models.py:278-515): no effect there
            l2_weight=float(hp.l2_regularization_weight) if (baseline and hp.use_l2_regularization) else 0.0,
            num_symbols=hp.num_symbols, embedding_dim=hp.embedding_dim,
            enc_prenet=tuple(hp.encoder_prenet_out_units), enc_prenet_drop=hp.encoder_prenet_drop_rate,
            conv_channels=hp.conv_channels, max_filter_width=hp.max_filter_width,
            proj1=hp.projection1_out_channels, proj2=hp.projection2_out_channels, num_highway=hp.num_highway,
            cbhg_out_units=hp.cbhg_out_units,
            sa_heads=hp.self_attention_num_heads, sa_drop=hp.self_attention_drop_rate,
            sa_num_hop=1 if baseline else int(hp.self_attention_num_hop),
            dec_sa_num_hop=1 if baseline else int(hp.decoder_self_attention_num_hop),
            dec_prenet=tuple(hp.decoder_prenet_out_units), dec_prenet_drop=hp.decoder_prenet_drop_rate,
            apply_dropout_on_inference=bool(hp.apply_dropout_on_inference),
            att_rnn_units=hp.attention_out_units, att_kernel=hp.attention_kernel, att_filters=hp.attention_filters,
            dec_units=hp.decoder_out_units,
            dec_sa_heads=hp.decoder_self_attention_num_heads, dec_sa_drop=hp.decoder_self_attention_drop_rate,
            num_mels=hp.num_mels, r=hp.outputs_per_step, n_feed_frame=hp.n_feed_frame,
            zc=hp.zoneout_factor_cell, zh=hp.zoneout_factor_output,
            num_speakers=hp.num_speakers if hp.use_speaker_embedding else 0, speaker_dim=hp.speaker_embedding_dim,
            speaker_offset=hp.speaker_embedding_offset,
            use_postnet_v2=bool(hp.use_postnet_v2), num_postnet_v2_layers=hp.num_postnet_v2_layers,
            postnet_v2_kernel_size=hp.postnet_v2_kernel_size, postnet_v2_out_channels=hp.postnet_v2_out_channels,
            postnet_v2_drop_rate=hp.postnet_v2_drop_rate)


def sa_prefix(base, hop):
    """parameter-name prefix of hop `hop` of a SelfAttentionTransformer stack: "enc.sa", "enc.sa.h1", "enc.sa.h2", ..."""
    return base if hop == 0 else "%s.h%d" % (base, hop)


def sa_prefixes(c):
    """[(prefix, units, heads)] of every SelfAttentionTransformer block of the configuration, encoder hops first"""
    out = [(sa_prefix("enc.sa", h), c.sa_units, c.sa_heads) for h in range(c.sa_num_hop if c.sa_units > 0 else 0)]
    return out + [(sa_prefix("dec.sa", h), c.dec_sa_units, c.dec_sa_heads) for h in range(c.dec_sa_num_hop if c.dec_sa_units > 0 else 0)]


def param_shapes(c):
    """Ordered (name, shape).  Encoder parameters first, decoder parameters after (two contiguous DP buckets)."""
    H = c.cbhg_out_units // 2
    L = [("embedding", (c.num_symbols, c.embedding_dim))]
    i = c.embedding_dim
    for n, o in enumerate(c.enc_prenet):
        L += [(f"enc.prenet{n}.W", (i, o)), (f"enc.prenet{n}.b", (o,))]
        i = o
    cin = c.enc_prenet[-1]
    for k in range(1, c.max_filter_width + 1):
        L.append((f"enc.bank{k}.W", (k, cin, c.conv_channels)))
    nb = c.max_filter_width * c.conv_channels
    L += [("enc.bank.gamma", (nb,)), ("enc.bank.beta", (nb,))]
    L += [("enc.proj1.W", (3, nb, c.proj1)), ("enc.proj1.gamma", (c.proj1,)), ("enc.proj1.beta", (c.proj1,))]
    L += [("enc.proj2.W", (3, c.proj1, c.proj2)), ("enc.proj2.gamma", (c.proj2,)), ("enc.proj2.beta", (c.proj2,))]
    for n in range(c.num_highway):
        L += [(f"enc.highway{n}.W", (H, 2 * H)), (f"enc.highway{n}.b", (2 * H,))]
    for d in ("fw", "bw"):
        L += [(f"enc.lstm_{d}.W", (2 * H, 4 * H)), (f"enc.lstm_{d}.b", (4 * H,))]
    S = c.sa_units
    if c.dual:
        L += [("enc.sa_proj.W", (c.cbhg_out_units, S)), ("enc.sa_proj.b", (S,))]
        for h in range(c.sa_num_hop):
            pre = sa_prefix("enc.sa", h)
            L += [(pre + ".kvq.W", (S, 3 * S)), (pre + ".kvq.b", (3 * S,)), (pre + ".o.W", (S, S)), (pre + ".o.b", (S,)),
                  (pre + ".t.W", (S, S)), (pre + ".t.b", (S,))]
    if c.num_speakers > 0:
        L.append(("speaker_embedding", (c.num_speakers, c.speaker_dim)))
    i = c.num_mels * c.n_feed_frame
    for n, o in enumerate(c.dec_prenet):
        L += [(f"dec.prenet{n}.W", (i, o)), (f"dec.prenet{n}.b", (o,))]
        i = o
    if c.num_speakers > 0:
        L += [("dec.prenet0.Ws", (c.speaker_dim, c.dec_prenet[0])), ("dec.prenet0.bs", (c.dec_prenet[0],)),
              ("dec.prenet0.W2", (c.dec_prenet[0], c.dec_prenet[0])), ("dec.prenet0.b2", (c.dec_prenet[0],))]
    A = c.att_rnn_units
    L += [("dec.att_lstm.W", (c.dec_prenet[-1] + c.ctx_dim + A, 4 * A)), ("dec.att_lstm.b", (4 * A,))]
    L += [("dec.att.Wq", (A, c.att1_units + c.att2_units)),
          ("dec.att1.Wm", (c.cbhg_out_units, c.att1_units)),
          ("dec.att1.F", (c.att_kernel, 1, c.att_filters)), ("dec.att1.bF", (c.att_filters,)),
          ("dec.att1.U", (c.att_filters, c.att1_units)), ("dec.att1.v", (c.att1_units,)),
          ("dec.att1.b", (c.att1_units,))]
    if c.transition_agent:
        L += [("dec.att1.Wa", (c.cbhg_out_units + c.att1_units, 1)), ("dec.att1.ba", (1,))]
    if c.dual:
        L += [("dec.att2.Wm", (c.sa_units, c.att2_units)), ("dec.att2.v", (c.att2_units,))]
    D = c.dec_units
    L += [("dec.lstm1.W", (A + c.ctx_dim + D, 4 * D)), ("dec.lstm1.b", (4 * D,))]
    L += [("dec.lstm2.W", (D + D, 4 * D)), ("dec.lstm2.b", (4 * D,))]
    S2 = c.dec_sa_units
    if S2 > 0:
        for h in range(c.dec_sa_num_hop):
            pre = sa_prefix("dec.sa", h)
            L += [(pre + ".kvq.W", (S2, 3 * S2)), (pre + ".kvq.b", (3 * S2,)), (pre + ".o.W", (S2, S2)),
                  (pre + ".o.b", (S2,)), (pre + ".t.W", (S2, S2)), (pre + ".t.b", (S2,))]
    L += [("dec.out.W", (c.out_in, c.num_mels * c.r + 1)), ("dec.out.b", (c.num_mels * c.r + 1,))]
    if c.use_postnet_v2:
        ci = c.num_mels
        for n in range(c.num_postnet_v2_layers):
            L += [(f"postnet.conv{n}.W", (c.postnet_v2_kernel_size, ci, c.postnet_v2_out_channels)),
                  (f"postnet.bn{n}.gamma", (c.postnet_v2_out_channels,)), (f"postnet.bn{n}.beta", (c.postnet_v2_out_channels,))]
            ci = c.postnet_v2_out_channels
        L += [("postnet.proj.W", (ci, c.num_mels)), ("postnet.proj.b", (c.num_mels,))]
    return L


def l2_regularized(c):
    """parameters that reference models/models.py:109-114 regularises: every trainable variable whose TF name contains none of
    "embedding", "bias", "batch_normalization", "output_projection_wrapper/kernel", "lstm_cell", the output / stop Dense of
    OutputAndStopTokenWrapper - i.e. the Dense / Conv1D kernels of pre-nets, conv bank, projections, highway layers and
    post-net, the attention's memory / query / location layers, location filter, `attention_variable` (v; `attention_bias` is
    a bias, modules/forward_attention.py:17-24) and the transition agent's kernel; NOT embeddings, biases, BatchNorm
    scale / offset, any LSTM kernel, the mel / stop projections."""
    out = []
    for name, _ in param_shapes(c):
        last = name.rsplit(".", 1)[-1]
        if name in ("embedding", "speaker_embedding") or last in ("b", "bs", "b2", "bF", "ba", "gamma", "beta"):
            continue
        if "lstm" in name or name.startswith("dec.out."):
            continue
        out.append(name)
    return out


def layout(c):
    """name -> (offset, shape); offsets padded to 8 floats: every fp32 view is 32-byte aligned and the bf16 shadows of
    the GEMM weights (same offsets, 2 bytes per element: engine.Engine.st_flat / sn_flat) are 16-byte aligned, which the
    large-tile kernels need for their vector loads."""
    off = 0
    out = {}
    for name, shp in param_shapes(c):
        n = int(np.prod(shp))
        out[name] = (off, shp)
        off += (n + 7) // 8 * 8
    return out, off


def init_params(c, seed=0):
    """Glorot-uniform weights, zero biases, highway transform bias -1, BN gamma 1."""
    g = np.random.default_rng(seed)
    P = {}
    for name, shp in param_shapes(c):
        last = name.rsplit(".", 1)[-1]
        if last == "gamma":
            a = np.ones(shp)
        elif last in ("beta", "b", "bs", "b2", "bF", "ba"):
            a = np.zeros(shp)
            if "highway" in name:
                a[shp[0] // 2:] = -1.0
        elif last == "v":
            lim = math.sqrt(6.0 / (shp[0] + 1))
            a = g.uniform(-lim, lim, shp)
        elif name in ("embedding", "speaker_embedding"):
            a = g.normal(0, 0.5, shp)
        else:
            fan_in, fan_out = (shp[0] * shp[1], shp[0] * shp[2]) if len(shp) == 3 else (shp[0], shp[1])
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            a = g.uniform(-lim, lim, shp)
        P[name] = a.astype(np.float32)
    return P


def to_reference_names(c):
    """Documentation of how the fused tensors map to the reference's tf variables (not used on the hot path)."""
    S, S2 = c.sa_units, c.dec_sa_units
    return {
        "enc.sa.kvq.W": ["key_projection/kernel [:, :%d]" % S, "value_projection/kernel [:, %d:%d]" % (S, 2 * S),
                         "query_projection/kernel [:, %d:]" % (2 * S)],
        "dec.sa.kvq.W": ["key_projection/kernel [:, :%d]" % S2, "value_projection/kernel", "query_projection/kernel"],
        "enc.highway{n}.W": ["H dense kernel [:, :H]", "T dense kernel [:, H:]"],
        "dec.att.Wq": ["ForwardAttention/query_layer/kernel [:, :%d]" % c.att1_units,
                       "BahdanauAttention/query_layer/kernel [:, %d:]" % c.att1_units],
        "dec.out.W": ["decoder/out_projection/kernel [:, :-1]", "decoder/stop_token_projection/kernel [:, -1:]"],
    }
