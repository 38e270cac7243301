"""TEST INFRASTRUCTURE ONLY (oracle): PyTorch-CPU restatement of the Self-attention Tacotron
teacher-forced training path.  *** parity unpinned *** — the reference (TF1 + external tacotron2@6af04c7)
cannot be imported or run here and ships no golden vectors; this file restates
SURVEY.md Appendix A and is cross-checked against the independent NumPy-float64 restatement in
oracle/numpy_ref.py.  It is never imported by the product path; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg use it.

Every function cites the reference file:line it follows (paths relative to /root/reference).
Parameters come in as a dict name -> torch tensor (any float dtype; float64 for tight checks).
Weight layouts: Dense W:[in,out]; Conv1D kernel [k,in,out]; LSTM kernel [in+h,4h] gate order i,j,f,o.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import rng


# ----------------------------------------------------------------------------------------------
# configuration (LJSpeech self-attention-tacotron.json resolved over hparams.py defaults)
# ----------------------------------------------------------------------------------------------
class Cfg:
    def __init__(self, **kw):
        self.num_symbols = 256; self.embedding_dim = 256
        self.enc_prenet = (256, 128); self.enc_prenet_drop = 0.5
        self.conv_channels = 128; self.max_filter_width = 16
        self.proj1 = 128; self.proj2 = 128; self.num_highway = 4; self.cbhg_out_units = 256
        self.sa_units = 32; self.sa_heads = 2; self.sa_drop = 0.05
        self.sa_num_hop = 1; self.dec_sa_num_hop = 1       # stacked SelfAttentionTransformer blocks (modules/module.py:411-419, :707-715)
        self.dec_prenet = (256, 128); self.dec_prenet_drop = 0.5; self.apply_dropout_on_inference = False
        self.att_rnn_units = 256; self.att1_units = 224; self.att2_units = 32
        self.att_kernel = 10; self.att_filters = 5
        # first-source mechanism (reference modules/attentions.py:25-62): "forward" or "location_sensitive";
        # cumulative_weights: the location features see the running SUM of the softmax alignments
        self.attention = "forward"; self.cumulative_weights = False
        # use_forward_attention_transition_agent (reference modules/forward_attention.py:80-86,111-116): the transition
        # probability u of the forward recursion is predicted per step instead of the constant 0.5
        self.transition_agent = False
        self.l2_weight = 0.0        # use_l2_regularization: baseline model_fn only (reference models/models.py:109-114)
        self.dec_units = 256; self.dec_sa_units = 256; self.dec_sa_heads = 2; self.dec_sa_drop = 0.05
        self.num_mels = 80; self.r = 2; self.n_feed_frame = 2
        self.zc = 0.1; self.zh = 0.1
        self.bn_eps = 1e-3; self.bn_momentum = 0.99
        self.num_speakers = 0; self.speaker_dim = 16; self.speaker_offset = 0
        self.conv_bias = False  # Conv1d believed bias-free (BN follows)
        # optional PostNetV2 (reference hparams.py:158-162; off in every shipped config)
        self.use_postnet_v2 = False; self.num_postnet_v2_layers = 5; self.postnet_v2_kernel_size = 5
        self.postnet_v2_out_channels = 512; self.postnet_v2_drop_rate = 0.5
        for k, v in kw.items():
            if not hasattr(self, k):
                raise KeyError(k)
            setattr(self, k, v)

    @property
    def ctx_dim(self):
        return self.cbhg_out_units + self.sa_units

    # sa_units = att2_units = 0 : single attention source (ZoneoutEncoderV1 + ExtendedDecoder's AttentionRNN, reference
    # modules/module.py:293-342,530-623); dec_sa_units = 0 : no decoder self-attention block, the projections read the
    # DecoderRNNV2 output directly (OutputAndStopTokenWrapper) - together the baseline ExtendedTacotronV1Model
    @property
    def dual(self):
        return self.sa_units > 0

    @property
    def out_in(self):
        return self.dec_sa_units if self.dec_sa_units > 0 else self.dec_units


def hop_prefix(base, hop):
    """parameters of hop `hop` of a SelfAttentionTransformer stack: base, base.h1, base.h2, ..."""
    return base if hop == 0 else "%s.h%d" % (base, hop)


HOP_STREAM = 64        # dropout stream of hop h of a stack = the stack's stream + HOP_STREAM * h


def param_shapes(cfg):
    """Ordered (name, shape) list — the build's own flat layout (product mirrors it in params.py)."""
    c = cfg
    H = c.cbhg_out_units // 2
    L = []
    L.append(("embedding", (c.num_symbols, c.embedding_dim)))
    i = c.embedding_dim
    for n, o in enumerate(c.enc_prenet):
        L += [(f"enc.prenet{n}.W", (i, o)), (f"enc.prenet{n}.b", (o,))]
        i = o
    cin = c.enc_prenet[-1]
    for k in range(1, c.max_filter_width + 1):
        L.append((f"enc.bank{k}.W", (k, cin, c.conv_channels)))
    L += [("enc.bank.gamma", (c.max_filter_width * c.conv_channels,)),
          ("enc.bank.beta", (c.max_filter_width * c.conv_channels,))]
    L += [("enc.proj1.W", (3, c.max_filter_width * c.conv_channels, c.proj1)),
          ("enc.proj1.gamma", (c.proj1,)), ("enc.proj1.beta", (c.proj1,))]
    L += [("enc.proj2.W", (3, c.proj1, c.proj2)), ("enc.proj2.gamma", (c.proj2,)), ("enc.proj2.beta", (c.proj2,))]
    for n in range(c.num_highway):
        L += [(f"enc.highway{n}.W", (H, 2 * H)), (f"enc.highway{n}.b", (2 * H,))]  # [H | T]
    for d in ("fw", "bw"):
        L += [(f"enc.lstm_{d}.W", (2 * H, 4 * H)), (f"enc.lstm_{d}.b", (4 * H,))]
    S = c.sa_units
    if c.dual:
        L += [("enc.sa_proj.W", (c.cbhg_out_units, S)), ("enc.sa_proj.b", (S,))]
        for h in range(c.sa_num_hop):
            pre = hop_prefix("enc.sa", h)
            L += [(pre + ".kvq.W", (S, 3 * S)), (pre + ".kvq.b", (3 * S,)),   # columns [K | V | Q]
                  (pre + ".o.W", (S, S)), (pre + ".o.b", (S,)),
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
    L += [("dec.att.Wq", (A, c.att1_units + c.att2_units)),          # columns [Wq1 | Wq2]
          ("dec.att1.Wm", (c.cbhg_out_units, c.att1_units)),
          ("dec.att1.F", (c.att_kernel, 1, c.att_filters)), ("dec.att1.bF", (c.att_filters,)),
          ("dec.att1.U", (c.att_filters, c.att1_units)), ("dec.att1.v", (c.att1_units,)),
          ("dec.att1.b", (c.att1_units,))]
    if c.transition_agent:       # Dense(1, sigmoid) on [context | processed query] (forward_attention.py:82-86,112-114)
        L += [("dec.att1.Wa", (c.cbhg_out_units + c.att1_units, 1)), ("dec.att1.ba", (1,))]
    if c.dual:
        L += [("dec.att2.Wm", (c.sa_units, c.att2_units)), ("dec.att2.v", (c.att2_units,))]
    D = c.dec_units
    L += [("dec.lstm1.W", (A + c.ctx_dim + D, 4 * D)), ("dec.lstm1.b", (4 * D,))]
    L += [("dec.lstm2.W", (D + D, 4 * D)), ("dec.lstm2.b", (4 * D,))]
    S2 = c.dec_sa_units
    if S2 > 0:
        for h in range(c.dec_sa_num_hop):
            pre = hop_prefix("dec.sa", h)
            L += [(pre + ".kvq.W", (S2, 3 * S2)), (pre + ".kvq.b", (3 * S2,)),
                  (pre + ".o.W", (S2, S2)), (pre + ".o.b", (S2,)),
                  (pre + ".t.W", (S2, S2)), (pre + ".t.b", (S2,))]
    L += [("dec.out.W", (c.out_in, c.num_mels * c.r + 1)), ("dec.out.b", (c.num_mels * c.r + 1,))]  # [mel(r*80) | stop]
    if c.use_postnet_v2:
        ci = c.num_mels
        for n in range(c.num_postnet_v2_layers):
            L += [(f"postnet.conv{n}.W", (c.postnet_v2_kernel_size, ci, c.postnet_v2_out_channels)),
                  (f"postnet.bn{n}.gamma", (c.postnet_v2_out_channels,)), (f"postnet.bn{n}.beta", (c.postnet_v2_out_channels,))]
            ci = c.postnet_v2_out_channels
        L += [("postnet.proj.W", (ci, c.num_mels)), ("postnet.proj.b", (c.num_mels,))]
    return L


def init_params(cfg, seed=0, dtype=np.float64):
    """Glorot-uniform weights, zero biases (highway T-bias -1, BN gamma 1) — SURVEY.md Appendix A."""
    g = np.random.default_rng(seed)
    P = {}
    for name, shp in param_shapes(cfg):
        if name.endswith(".gamma"):
            a = np.ones(shp)
        elif name.rsplit(".", 1)[-1] in ("beta", "b", "bs", "b2", "bF"):
            a = np.zeros(shp)
            if "highway" in name:
                a[shp[0] // 2:] = -1.0
        elif name.endswith(".v"):
            lim = math.sqrt(6.0 / (shp[0] + 1))
            a = g.uniform(-lim, lim, shp)
        elif name in ("embedding", "speaker_embedding"):
            a = g.normal(0, 0.5, shp)
        else:
            if len(shp) == 3:
                fan_in, fan_out = shp[0] * shp[1], shp[0] * shp[2]
            else:
                fan_in, fan_out = shp[0], shp[1]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            a = g.uniform(-lim, lim, shp)
        P[name] = a.astype(dtype)
    return P


# ----------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------
def _mask(seed, stream, shape, rate, like):
    return torch.from_numpy(rng.keep_mask(seed, stream, shape, rate)).to(like.dtype)


def dropout(x, rate, training, seed, stream):
    """tf.layers.dropout: kept units scaled by 1/(1-rate).  Mask from oracle/rng.py; idx = C-order index."""
    if not training or rate <= 0.0:
        return x
    return x * _mask(seed, stream, tuple(x.shape), rate, x) / (1.0 - rate)


def prenet(x, P, prefix, n_layers, rate, training, seed, streams, speaker_embed=None, on_inference=False, step=None):
    """PreNet (external tacotron2) / MultiSpeakerPreNet for layer 0 when speaker_embed is given
    (reference modules/multi_speaker_modules.py:27-32).  on_inference: apply_dropout_on_inference - the plain PreNet layers
    (not MultiSpeakerPreNet: reference modules/module.py:569-577) keep their dropout when training is False.
    step=(t, T): x is row t of a [B, T, .] sequence (step-by-step decode) - the mask is the one of that row."""
    def drop(y, on, stream):
        if step is None or not on or rate <= 0.0:
            return dropout(y, rate, on, seed, stream)
        t, T = step
        Bn, N = y.shape
        m = _mask(seed, stream, (Bn, T, N), rate, y)[:, t]
        return y * m / (1.0 - rate)
    for n in range(n_layers):
        y = x @ P[f"{prefix}{n}.W"] + P[f"{prefix}{n}.b"]
        if n == 0 and speaker_embed is not None:
            # dense0 = relu(xW0+b0) + softsign(s Ws+bs); dense = relu(dense0 W2+b2); dropout  (reference :27-32)
            s = speaker_embed @ P[f"{prefix}0.Ws"] + P[f"{prefix}0.bs"]
            s = s / (1.0 + s.abs())
            y = torch.relu(y) + (s[:, None, :] if y.dim() == 3 else s)
            y = torch.relu(y @ P[f"{prefix}0.W2"] + P[f"{prefix}0.b2"])
        else:
            y = torch.relu(y)
        plain = not (n == 0 and speaker_embed is not None)
        x = drop(y, training or (on_inference and plain), streams[n])
    return x


def conv1d_same(x, W, b=None):
    """tf.layers.Conv1D(padding='SAME'): pad_left=(k-1)//2, pad_right=(k-1)-pad_left.
    x [B,T,Cin], W [k,Cin,Cout]."""
    k = W.shape[0]
    pl = (k - 1) // 2
    pr = (k - 1) - pl
    xp = F.pad(x, (0, 0, pl, pr))
    T = x.shape[1]
    y = 0
    for j in range(k):
        y = y + xp[:, j:j + T, :] @ W[j]
    if b is not None:
        y = y + b
    return y


def batch_norm(x, gamma, beta, eps, training, moving=None):
    """tf.layers.batch_normalization over the channel axis; training statistics over ALL B*T positions incl.
    padding (SURVEY.md A.3, §7 'BatchNorm over padded batches')."""
    if training:
        mean = x.mean(dim=(0, 1))
        var = ((x - mean) ** 2).mean(dim=(0, 1))
    else:
        mean, var = moving
    return (x - mean) / torch.sqrt(var + eps) * gamma + beta


def lstm_cell(x, c, h, W, b):
    """tf.nn.rnn_cell.LSTMCell, forget_bias=1.0, gate order i,j,f,o."""
    z = torch.cat([x, h], dim=-1) @ W + b
    n = c.shape[-1]
    i, j, f, o = z[..., :n], z[..., n:2 * n], z[..., 2 * n:3 * n], z[..., 3 * n:]
    c_new = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
    h_new = torch.sigmoid(o) * torch.tanh(c_new)
    return c_new, h_new


def zoneout(new, old, rate, training, keep):
    """ZoneoutLSTMCell (external tacotron2). training: per-element keep-new with prob 1-rate;
    inference: interpolation."""
    if training:
        if rate <= 0.0:
            return new
        return torch.where(keep, new, old)
    return (1.0 - rate) * new + rate * old


def _zmask(seed, stream, B, T, t, H, rate, training):
    """zoneout keep mask for time step t: idx = (b*T + t)*H + j."""
    if not training or rate <= 0.0:
        return None
    b = np.arange(B, dtype=np.uint64)[:, None]
    j = np.arange(H, dtype=np.uint64)[None, :]
    idx = (b * np.uint64(T) + np.uint64(t)) * np.uint64(H) + j
    return torch.from_numpy(rng.hash_u32(seed, stream, idx) >= rng.rate_threshold(rate))


def zoneout_lstm_seq(x, W, b, H, lengths, reverse, zc, zh, training, seed, streams):
    """One direction of tf.nn.bidirectional_dynamic_rnn over ZoneoutLSTMCell with sequence_length
    (reference modules/module.py:93-108): outputs beyond length are zero, state frozen;
    backward direction walks t = len-1 .. 0.  Cell output = h' (pre-zoneout), carried state = zoneout."""
    B, T, _ = x.shape
    c = x.new_zeros(B, H); h = x.new_zeros(B, H)
    out = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        c_new, h_new = lstm_cell(x[:, t], c, h, W, b)
        kc = _zmask(seed, streams[0], B, T, t, H, zc, training)
        kh = _zmask(seed, streams[1], B, T, t, H, zh, training)
        c_out = zoneout(c_new, c, zc, training, kc)
        h_out = zoneout(h_new, h, zh, training, kh)
        if lengths is not None:
            valid = (t < lengths)[:, None]
            out[t] = torch.where(valid, h_new, torch.zeros_like(h_new))
            c = torch.where(valid, c_out, c); h = torch.where(valid, h_out, h)
        else:
            out[t] = h_new
            c, h = c_out, h_out
    return torch.stack(out, dim=1)


def sdpa(q, k, v, causal, rate, training, seed, stream):
    """ScaledDotProductAttentionMechanism.__call__ (reference modules/self_attention.py:45-65): no padding mask
    (use_padding_mask=False everywhere), optional subsequent mask (:79-86), dropout on probs (:61),
    returns pre-dropout probs as alignments (:59).  q,k,v [B,h,T,hd]."""
    hd = q.shape[-1]
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if causal:
        T = s.shape[-1]
        m = torch.tril(torch.ones(T, T, dtype=torch.bool))
        s = torch.where(m, s, torch.full_like(s, float("-inf")))
    p = torch.softmax(s, dim=-1)
    pd = dropout(p, rate, training, seed, stream)
    return pd @ v, p


def self_attention_transformer(x, P, prefix, heads, causal, rate, training, seed, stream):
    """SelfAttentionTransformer.call (reference modules/module.py:363-371) over MultiHeadAttention.call
    (modules/self_attention.py:108-128): x + tanh(Dense(MHA(x)))."""
    B, T, D = x.shape
    kvq = x @ P[f"{prefix}.kvq.W"] + P[f"{prefix}.kvq.b"]
    k, v, q = kvq[..., :D], kvq[..., D:2 * D], kvq[..., 2 * D:]
    sp = lambda t: t.reshape(B, T, heads, D // heads).permute(0, 2, 1, 3)
    o, align = sdpa(sp(q), sp(k), sp(v), causal, rate, training, seed, stream)
    o = o.permute(0, 2, 1, 3).reshape(B, T, D)
    o = o @ P[f"{prefix}.o.W"] + P[f"{prefix}.o.b"]
    tr = torch.tanh(o @ P[f"{prefix}.t.W"] + P[f"{prefix}.t.b"])
    return x + tr, align


# ----------------------------------------------------------------------------------------------
# encoder  (reference modules/module.py:30-113, 374-441)
# ----------------------------------------------------------------------------------------------
def transformer_stack(x, P, base, num_hop, heads, causal, rate, training, seed, stream, collect=None, key=None):
    """num_hop SelfAttentionTransformer blocks applied in sequence, each with its own weights (reference
    modules/module.py:411-419 + :433-439 encoder, :707-715 + :753-757 decoder: reduce over the list, alignments of all hops
    collected).  Returns (output, alignments of the FIRST hop); collect[key] = the alignments of every hop."""
    aligns = []
    for h in range(num_hop):
        x, a = self_attention_transformer(x, P, hop_prefix(base, h), heads, causal, rate, training, seed, stream + HOP_STREAM * h)
        aligns.append(a)
    if collect is not None and key is not None:
        collect[key] = aligns
    return x, aligns[0]


def encoder(source, source_length, P, cfg, training, seed, bn_moving=None, collect=None):
    emb = P["embedding"][source]                                   # models/models.py:351 (A.1)
    x = prenet(emb, P, "enc.prenet", len(cfg.enc_prenet), cfg.enc_prenet_drop, training, seed,
               (rng.STREAM_ENC_PRENET0, rng.STREAM_ENC_PRENET1))   # module.py:426
    # ZoneoutCBHG.call module.py:77-110
    bank = torch.cat([conv1d_same(x, P[f"enc.bank{k}.W"]) for k in range(1, cfg.max_filter_width + 1)], dim=-1)
    mv = (lambda n: None) if bn_moving is None else (lambda n: bn_moving[n])
    bank = torch.relu(batch_norm(bank, P["enc.bank.gamma"], P["enc.bank.beta"], cfg.bn_eps, training, mv("bank")))
    nxt = torch.cat([bank[:, 1:], bank[:, -1:]], dim=1)            # MaxPooling1D(2,1,'SAME'): M[T-1]=C[T-1]
    mp = torch.maximum(bank, nxt)
    p1 = torch.relu(batch_norm(conv1d_same(mp, P["enc.proj1.W"]), P["enc.proj1.gamma"], P["enc.proj1.beta"],
                               cfg.bn_eps, training, mv("proj1")))
    p2 = batch_norm(conv1d_same(p1, P["enc.proj2.W"]), P["enc.proj2.gamma"], P["enc.proj2.beta"],
                    cfg.bn_eps, training, mv("proj2"))
    hw = p2 + x                                                    # module.py:86 residual
    H = cfg.cbhg_out_units // 2
    for n in range(cfg.num_highway):                               # HighwayNet (A.5)
        z = hw @ P[f"enc.highway{n}.W"] + P[f"enc.highway{n}.b"]
        hh, tt = torch.relu(z[..., :H]), torch.sigmoid(z[..., H:])
        hw = hh * tt + hw * (1.0 - tt)
    fw = zoneout_lstm_seq(hw, P["enc.lstm_fw.W"], P["enc.lstm_fw.b"], H, source_length, False, cfg.zc, cfg.zh,
                          training, seed, (rng.STREAM_ENC_LSTM_FW_C, rng.STREAM_ENC_LSTM_FW_H))
    bw = zoneout_lstm_seq(hw, P["enc.lstm_bw.W"], P["enc.lstm_bw.b"], H, source_length, True, cfg.zc, cfg.zh,
                          training, seed, (rng.STREAM_ENC_LSTM_BW_C, rng.STREAM_ENC_LSTM_BW_H))
    lstm_out = torch.cat([fw, bw], dim=-1)                         # module.py:110
    if not cfg.dual:                                               # ZoneoutEncoderV1.call (module.py:336-339): CBHG only
        if collect is not None:
            collect.update(emb=emb, prenet=x, bank=bank, maxpool=mp, proj1=p1, proj2=p2, highway=hw)
        return lstm_out, None, None
    sa_in = lstm_out @ P["enc.sa_proj.W"] + P["enc.sa_proj.b"]     # module.py:429
    sa_out, align = transformer_stack(sa_in, P, "enc.sa", cfg.sa_num_hop, cfg.sa_heads, False, cfg.sa_drop, training,
                                      seed, rng.STREAM_ENC_SA, collect, "enc_alignments")
    if collect is not None:
        collect.update(emb=emb, prenet=x, bank=bank, maxpool=mp, proj1=p1, proj2=p2, highway=hw, sa_in=sa_in)
    return lstm_out, sa_out, align


# ----------------------------------------------------------------------------------------------
# decoder  (reference modules/module.py:1011-1042,1449-1559 ; forward_attention.py )
# ----------------------------------------------------------------------------------------------
def masked_softmax(e, lengths):
    """TF _maybe_mask_score(-inf) then softmax."""
    T = e.shape[1]
    m = torch.arange(T)[None, :] < lengths[:, None]
    e = torch.where(m, e, torch.full_like(e, float("-inf")))
    return torch.softmax(e, dim=-1)


def forward_attention_step(query, keys, state, P, lengths, mode="forward", cumulative=False, values=None, agent=False):
    """ForwardAttention.__call__ (reference modules/forward_attention.py:88-122; no transition agent) and, with
    mode="location_sensitive", tacotron2's LocationSensitiveAttention (external, SURVEY.md 8c: the same
    _location_sensitive_score :13-26, alignments = softmax(energy), no alpha recursion).  cumulative: the next state's
    location-conv input is alignments + previous_alignments (:118-119) instead of the alignments (:120-121)."""
    a_prev, alpha_prev, u = state
    pq = query @ P["dec.att.Wq"][:, :keys.shape[-1]]                # :92 query_layer (no bias)
    f = conv1d_same(a_prev[:, :, None], P["dec.att1.F"], P["dec.att1.bF"])   # :98-100
    lf = f @ P["dec.att1.U"]                                        # :101
    e = (P["dec.att1.v"] * torch.tanh(keys + pq[:, None, :] + lf + P["dec.att1.b"])).sum(-1)   # :26
    a = masked_softmax(e, lengths)                                  # :105 _probability_fn
    nxt = a + a_prev if cumulative else a
    if mode == "location_sensitive":
        return a, (nxt, alpha_prev, u)
    shifted = F.pad(alpha_prev[:, :-1], (1, 0))                     # :108
    alpha = ((1 - u) * alpha_prev + u * shifted + 1e-7) * a         # :109
    alpha_n = alpha / alpha.sum(dim=1, keepdim=True)                # :110
    if agent:                                                       # :111-114 transition agent: u_{t+1} from this step
        ctx = (alpha_n[:, :, None] * values).sum(1)                 # _calculate_context (:28-40)
        u = torch.sigmoid(torch.cat([ctx, pq], dim=-1) @ P["dec.att1.Wa"] + P["dec.att1.ba"])    # [B, 1]
    return alpha_n, (nxt, alpha_n, u)                               # :118-121


def additive_attention_step(query, keys, P, lengths):
    """tf.contrib.seq2seq.BahdanauAttention, normalize=False (reference modules/attentions.py:53-57; A.8)."""
    pq = query @ P["dec.att.Wq"][:, -keys.shape[-1]:]
    e = (P["dec.att2.v"] * torch.tanh(keys + pq[:, None, :])).sum(-1)
    return masked_softmax(e, lengths)


def decoder_rnn(lstm_out, sa_out, source_length, target, P, cfg, training, seed, speaker_embed=None,
                collect=None):
    """Teacher-forced dynamic_decode over DecoderRNNV2(DualSourceAttentionRNN, ZoneoutLSTM, ZoneoutLSTM)
    fed by TransformerTrainingHelper (reference modules/helpers.py:13-55; module.py:1018-1025,1516-1534).
    Returns decoder outputs [B,Td,256] and the two alignment histories [B,Td,Ti]."""
    B, Tm, _ = target.shape
    r = cfg.r
    Td = Tm // r
    Ti = lstm_out.shape[1]
    tg = target.reshape(B, Td, cfg.num_mels * r)                    # helpers.py:22-24
    feed = cfg.num_mels * cfg.n_feed_frame
    go = target.new_zeros(B, 1, feed)                               # helpers.py:42-45,224-225
    dec_in = torch.cat([go, tg[:, :-1, -feed:]], dim=1)             # helpers.py:51-55
    pre = prenet(dec_in, P, "dec.prenet", len(cfg.dec_prenet), cfg.dec_prenet_drop, training, seed,
                 (rng.STREAM_DEC_PRENET0, rng.STREAM_DEC_PRENET1), speaker_embed, on_inference=cfg.apply_dropout_on_inference)
    # memories: values = memory * seq_mask ; keys = values W_m  (BahdanauAttention._prepare_memory; A.7)
    mm = (torch.arange(Ti)[None, :] < source_length[:, None]).to(lstm_out.dtype)[:, :, None]
    values1 = lstm_out * mm
    keys1 = values1 @ P["dec.att1.Wm"]
    values2 = keys2 = None
    if cfg.dual:
        values2 = sa_out * mm
        keys2 = values2 @ P["dec.att2.Wm"]
    A, D = cfg.att_rnn_units, cfg.dec_units
    c0 = target.new_zeros(B, A); h0 = target.new_zeros(B, A)
    c1 = target.new_zeros(B, D); h1 = target.new_zeros(B, D)
    c2 = target.new_zeros(B, D); h2 = target.new_zeros(B, D)
    attn = target.new_zeros(B, cfg.ctx_dim)
    a_prev = target.new_zeros(B, Ti)
    alpha_prev = torch.cat([target.new_ones(B, 1), target.new_zeros(B, Ti - 1)], dim=1)   # forward_attention.py:128-136
    st1 = (a_prev, alpha_prev, 0.5)
    outs, al1, al2, att_out = [], [], [], []
    for t in range(Td):
        # AttentionWrapper step (A.9): cell_in = concat(prenet(x_t), attention_{t-1})
        cin = torch.cat([pre[:, t], attn], dim=-1)
        cn, hn = lstm_cell(cin, c0, h0, P["dec.att_lstm.W"], P["dec.att_lstm.b"])
        c0 = zoneout(cn, c0, cfg.zc, training, _zmask(seed, rng.STREAM_ATT_LSTM_C, B, Td, t, A, cfg.zc, training))
        h0 = zoneout(hn, h0, cfg.zh, training, _zmask(seed, rng.STREAM_ATT_LSTM_H, B, Td, t, A, cfg.zh, training))
        query = hn                                                  # pre-zoneout cell output
        alpha, st1 = forward_attention_step(query, keys1, st1, P, source_length, cfg.attention, cfg.cumulative_weights,
                                            values1, cfg.transition_agent)
        ctx1 = (alpha[:, :, None] * values1).sum(1)
        if cfg.dual:
            a2 = additive_attention_step(query, keys2, P, source_length)
            ctx2 = (a2[:, :, None] * values2).sum(1)
            attn = torch.cat([ctx1, ctx2], dim=-1)
        else:                                                       # AttentionRNN: one mechanism (module.py:566-574)
            a2 = torch.zeros_like(alpha)
            attn = ctx1
        x1 = torch.cat([hn, attn], dim=-1)                          # ConcatOutputAndAttentionWrapper
        att_out.append(x1)
        cn1, hn1 = lstm_cell(x1, c1, h1, P["dec.lstm1.W"], P["dec.lstm1.b"])
        c1 = zoneout(cn1, c1, cfg.zc, training, _zmask(seed, rng.STREAM_LSTM1_C, B, Td, t, D, cfg.zc, training))
        h1 = zoneout(hn1, h1, cfg.zh, training, _zmask(seed, rng.STREAM_LSTM1_H, B, Td, t, D, cfg.zh, training))
        cn2, hn2 = lstm_cell(hn1, c2, h2, P["dec.lstm2.W"], P["dec.lstm2.b"])
        c2 = zoneout(cn2, c2, cfg.zc, training, _zmask(seed, rng.STREAM_LSTM2_C, B, Td, t, D, cfg.zc, training))
        h2 = zoneout(hn2, h2, cfg.zh, training, _zmask(seed, rng.STREAM_LSTM2_H, B, Td, t, D, cfg.zh, training))
        outs.append(hn2); al1.append(alpha); al2.append(a2)
    if collect is not None:
        collect.update(dec_prenet=pre, keys1=keys1, values1=values1, keys2=keys2, values2=values2,
                       att_out=torch.stack(att_out, 1))
    return torch.stack(outs, 1), torch.stack(al1, 1), torch.stack(al2, 1)


def decoder(lstm_out, sa_out, source_length, target, P, cfg, training, seed, speaker_embed=None, collect=None):
    """DualSourceTransformerDecoder.call training branch (reference modules/module.py:1493-1559) +
    RNNTransformer.__call__ training branch (:741-760)."""
    dec_out, al1, al2 = decoder_rnn(lstm_out, sa_out, source_length, target, P, cfg, training, seed,
                                    speaker_embed, collect)
    if cfg.dec_sa_units > 0:
        tr, dec_align = transformer_stack(dec_out, P, "dec.sa", cfg.dec_sa_num_hop, cfg.dec_sa_heads, True, cfg.dec_sa_drop,
                                          training, seed, rng.STREAM_DEC_SA, collect, "dec_alignments")
    else:                          # ExtendedDecoder (module.py:588-590): OutputAndStopTokenWrapper on the RNN output
        tr, dec_align = dec_out, None
    y = tr @ P["dec.out.W"] + P["dec.out.b"]                        # Projection (module.py:626-643)
    B, Td, _ = y.shape
    mel = y[..., :-1].reshape(B, Td * cfg.r, cfg.num_mels)          # module.py:1558
    stop = y[..., -1:]
    if collect is not None:
        collect.update(dec_out=dec_out, transformed=tr)
    return mel, stop, al1, al2, dec_align


def infer(P, source, source_length, cfg, max_steps, bn_moving, speaker_id=None, teacher=None, min_steps=10,
          stop_threshold=0.5, teacher_alignments=None, seed=0):
    """Step-by-step decode with is_training=False: RNNTransformer else-branch (reference
    modules/module.py:762-778) = per step, DecoderRNNV2 cell, the decoder output appended to a history
    (RNNStateHistoryWrapper, modules/rnn_wrappers.py:47-80), the causal SelfAttentionTransformer re-run over the WHOLE
    history and its last row projected (TransformerWrapper :87-124, OutputAndStopTokenTransparentWrapper :188-214).
    teacher=None: free running (StopTokenBasedInferenceHelper: next input = last n_feed_frame*num_mels outputs; stop
    when sigmoid(stop) > threshold for every sample and t > min_steps, or at max_steps).
    teacher=[B,Tm,num_mels]: ValidationHelper semantics (inputs from the ground truth; exactly Tm/r steps) - the
    reference's own test property says this equals the batched training-branch output (transformer_test.py:40-82).
    teacher_alignments=(a1, a2) [B,T,Ti]: forced-alignment mode (reference modules/teacher_forcing_attention.py:13-78,
    models/models.py:411-428): both mechanisms return the given alignment of the step instead of their own.
    Zoneout in interpolation mode, dropout off - except the plain decoder PreNet layers under apply_dropout_on_inference
    (mask of row (b, t) of a [B, max_steps, units] activation from `seed`) -, BatchNorm on moving statistics."""
    training = False
    spk = None
    if cfg.num_speakers > 0:
        spk = P["speaker_embedding"][speaker_id - cfg.speaker_offset]
    lstm_out, sa_out, enc_align = encoder(source, source_length, P, cfg, training, seed, bn_moving=bn_moving)
    B, Ti = source.shape
    r, nm = cfg.r, cfg.num_mels
    feed = nm * cfg.n_feed_frame
    if teacher is not None:
        max_steps = teacher.shape[1] // r
        tg = teacher.reshape(B, max_steps, nm * r)
    mm = (torch.arange(Ti)[None, :] < source_length[:, None]).to(lstm_out.dtype)[:, :, None]
    values1 = lstm_out * mm; keys1 = values1 @ P["dec.att1.Wm"]
    if cfg.dual:
        values2 = sa_out * mm; keys2 = values2 @ P["dec.att2.Wm"]
    A, D = cfg.att_rnn_units, cfg.dec_units
    z = lambda n: lstm_out.new_zeros(B, n)
    c0, h0, c1, h1, c2, h2, attn = z(A), z(A), z(D), z(D), z(D), z(D), z(cfg.ctx_dim)
    st1 = (z(Ti), torch.cat([lstm_out.new_ones(B, 1), z(Ti - 1)], dim=1), 0.5)
    x_in = z(feed)                                                   # go frame
    hist, mels, stops, al1, al2 = [], [], [], [], []
    for t in range(max_steps):
        pre = prenet(x_in, P, "dec.prenet", len(cfg.dec_prenet), cfg.dec_prenet_drop, training, seed,
                     (rng.STREAM_DEC_PRENET0, rng.STREAM_DEC_PRENET1), spk, on_inference=cfg.apply_dropout_on_inference,
                     step=(t, max_steps))
        cn, hn = lstm_cell(torch.cat([pre, attn], dim=-1), c0, h0, P["dec.att_lstm.W"], P["dec.att_lstm.b"])
        c0 = zoneout(cn, c0, cfg.zc, training, None); h0 = zoneout(hn, h0, cfg.zh, training, None)
        if teacher_alignments is not None:      # TeacherForcing*Attention.__call__: alignments = teacher[:, index]
            alpha, a2 = teacher_alignments[0][:, t], teacher_alignments[1][:, t]
        else:
            alpha, st1 = forward_attention_step(hn, keys1, st1, P, source_length, cfg.attention, cfg.cumulative_weights,
                                                values1, cfg.transition_agent)
            a2 = additive_attention_step(hn, keys2, P, source_length) if cfg.dual else torch.zeros_like(alpha)
        attn = (alpha[:, :, None] * values1).sum(1)
        if cfg.dual:
            attn = torch.cat([attn, (a2[:, :, None] * values2).sum(1)], dim=-1)
        cn1, hn1 = lstm_cell(torch.cat([hn, attn], dim=-1), c1, h1, P["dec.lstm1.W"], P["dec.lstm1.b"])
        c1 = zoneout(cn1, c1, cfg.zc, training, None); h1 = zoneout(hn1, h1, cfg.zh, training, None)
        cn2, hn2 = lstm_cell(hn1, c2, h2, P["dec.lstm2.W"], P["dec.lstm2.b"])
        c2 = zoneout(cn2, c2, cfg.zc, training, None); h2 = zoneout(hn2, h2, cfg.zh, training, None)
        hist.append(hn2)
        if cfg.dec_sa_units > 0:
            tr, _ = transformer_stack(torch.stack(hist, 1), P, "dec.sa", cfg.dec_sa_num_hop, cfg.dec_sa_heads, True,
                                      cfg.dec_sa_drop, training, seed, rng.STREAM_DEC_SA)      # whole history, last row used
            last = tr[:, -1]
        else:
            last = hn2
        y = last @ P["dec.out.W"] + P["dec.out.b"]
        mels.append(y[:, :-1]); stops.append(y[:, -1]); al1.append(alpha); al2.append(a2)
        if teacher is not None:
            x_in = tg[:, t, nm * r - feed:]
        else:
            x_in = y[:, nm * r - feed:nm * r]
            if t > min_steps and bool((torch.sigmoid(y[:, -1]) > stop_threshold).all()):
                break
    T = len(mels)
    return dict(mel=torch.stack(mels, 1).reshape(B, T * r, nm), stop=torch.stack(stops, 1)[..., None],
                alignment1=torch.stack(al1, 1), alignment2=torch.stack(al2, 1), steps=T,
                lstm_out=lstm_out, sa_out=sa_out)


# ----------------------------------------------------------------------------------------------
# loss + optimiser (reference models/models.py:467-498,594-598 )
# ----------------------------------------------------------------------------------------------
def losses(mel, stop, batch, loss_type="l1"):
    d = mel - batch["mel"]
    e = d.abs() if loss_type == "l1" else d * d
    w = batch["spec_loss_mask"][:, :, None]
    mel_loss = (e * w).sum() / (mel.shape[-1] * batch["spec_loss_mask"].sum())
    x = stop[..., 0]
    z = batch["done"]
    bce = torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs()))
    done_loss = (bce * batch["binary_loss_mask"]).sum() / batch["binary_loss_mask"].sum()
    return mel_loss, done_loss


def postnet_v2(mel, P, cfg, training, seed, bn_moving=None):
    """PostNetV2 (external tacotron2; call site reference models/models.py:92-100): num_layers x
    [Conv1d(k) -> BN -> tanh (last layer: linear) -> dropout], Dense(out_channels -> num_mels), residual add."""
    x = mel
    L = cfg.num_postnet_v2_layers
    mv = (lambda n: None) if bn_moving is None else (lambda n: bn_moving[n])
    for n in range(L):
        y = batch_norm(conv1d_same(x, P[f"postnet.conv{n}.W"]), P[f"postnet.bn{n}.gamma"], P[f"postnet.bn{n}.beta"],
                       cfg.bn_eps, training, mv(f"postnet{n}"))
        if n < L - 1:
            y = torch.tanh(y)
        x = dropout(y, cfg.postnet_v2_drop_rate, training, seed, rng.STREAM_POSTNET0 + n)
    return mel + (x @ P["postnet.proj.W"] + P["postnet.proj.b"])


def forward(P, batch, cfg, training=True, seed=0, collect=None):
    """model_fn forward, TRAIN mode (reference models/models.py:278-482)."""
    spk = None
    if cfg.num_speakers > 0:
        spk = P["speaker_embedding"][batch["speaker_id"] - cfg.speaker_offset]
    lstm_out, sa_out, enc_align = encoder(batch["source"], batch["source_length"], P, cfg, training, seed,
                                          collect=collect)
    mel, stop, al1, al2, dec_align = decoder(lstm_out, sa_out, batch["source_length"], batch["mel"], P, cfg,
                                             training, seed, spk, collect)
    mel_loss, done_loss = losses(mel, stop, batch)
    out = dict(mel=mel, stop=stop, alignment1=al1, alignment2=al2, enc_alignment=enc_align,
               dec_alignment=dec_align, lstm_out=lstm_out, sa_out=sa_out,
               mel_loss=mel_loss, done_loss=done_loss, loss=mel_loss + done_loss)
    if cfg.use_postnet_v2:                   # extra spec_loss term (reference models/models.py:116-118)
        post = postnet_v2(mel, P, cfg, training, seed)
        pl, _ = losses(post, stop, batch)
        out.update(mel_postnet=post, postnet_mel_loss=pl, loss=out["loss"] + pl)
    if cfg.l2_weight > 0 and training:       # modules/regularizers.py:11-18 with the blacklist of models/models.py:109-111
        reg = cfg.l2_weight * sum(0.5 * (P[k] ** 2).sum() for k in l2_regularized(P))
        out.update(regularization_loss=reg, loss=out["loss"] + reg)
    return out


def l2_regularized(P):
    """names the blacklist (embedding, bias, batch_normalization, lstm_cell, output / stop Dense; models/models.py:109-111)
    leaves: Dense / Conv1D kernels, the attention's layers, filter and attention_variable, the agent's kernel"""
    out = []
    for k in P:
        last = k.rsplit(".", 1)[-1]
        if k in ("embedding", "speaker_embedding") or last in ("b", "bs", "b2", "bF", "ba", "gamma", "beta"):
            continue
        if "lstm" in k or k.startswith("dec.out."):
            continue
        out.append(k)
    return out


def learning_rate(init_rate, global_step, step_factor=1, warmup=4000.0):
    """models/models.py:594-598."""
    s = float(global_step * step_factor + 1)
    return init_rate * warmup ** 0.5 * min(s * warmup ** -1.5, s ** -0.5)


def clip_and_adam(params, grads, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8, clip=1.0):
    """tf.clip_by_global_norm(…,1.0) + tf.train.AdamOptimizer (models/models.py:489-498; SURVEY.md A.11:
    epsilon on the UNcorrected sqrt(v)).  In-place on dicts of torch tensors; t = 1-based step."""
    gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    scale = 1.0 / max(1.0, gn / clip)
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    for k in params:
        g = grads[k] * scale
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).add_(g * g, alpha=1 - b2)
        params[k].sub_(lr_t * m[k] / (v[k].sqrt() + eps))
    return gn


def to_torch(P, dtype=torch.float64, requires_grad=False):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad) for k, v in P.items()}


def batch_to_torch(batch, dtype=torch.float64):
    out = {}
    for k, v in batch.items():
        a = np.asarray(v)
        out[k] = torch.tensor(a, dtype=torch.long if a.dtype.kind in "iu" else dtype)
    return out
