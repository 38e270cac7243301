"""TEST INFRASTRUCTURE ONLY (oracle): independent NumPy-float64 restatement of the forward pass of the
Self-attention Tacotron teacher-forced training path (no torch, no autograd; explicit per-sample / per-step
loops).  *** parity unpinned *** (SURVEY.md §8c): TF1 and tacotron2@6af04c7f6d7ad212e499bbc671802acfbcbe2404 are
not installable here and the reference ships no golden vectors; this file follows SURVEY.md Appendix A and
the in-tree reference lines cited per function (paths relative to /root/reference).  It generates the fixtures
under tests/golden/ (tests/golden/make_golden.py) and cross-checks oracle/torch_ref.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
import numpy as np

from . import rng


HOP_STREAM = 64        # dropout stream of hop h of a SelfAttentionTransformer stack = the stack's stream + HOP_STREAM * h


def hop_prefix(base, hop):
    return base if hop == 0 else "%s.h%d" % (base, hop)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def keep(seed, stream, idx, rate):
    return rng.hash_u32(seed, stream, idx) >= rng.rate_threshold(rate)


def dense(x, W, b=None):
    y = x @ W
    return y if b is None else y + b


def dropout_rows(y, rate, training, seed, stream, row0=0):
    """y [rows, C]; element index = (row0+row)*C + c (C-order over the full [B*T, C] activation)."""
    if not training or rate <= 0:
        return y
    R, C = y.shape
    idx = (np.arange(R, dtype=np.uint64)[:, None] + np.uint64(row0)) * np.uint64(C) + np.arange(C, dtype=np.uint64)
    return np.where(keep(seed, stream, idx, rate), y / (1.0 - rate), 0.0)


def conv1d_same_sample(x, W):
    """x [T,Cin], W [k,Cin,Cout]; SAME: pad_left=(k-1)//2."""
    k = W.shape[0]
    T = x.shape[0]
    pl = (k - 1) // 2
    y = np.zeros((T, W.shape[2]))
    for t in range(T):
        for j in range(k):
            s = t + j - pl
            if 0 <= s < T:
                y[t] += x[s] @ W[j]
    return y


def bn_train(x, gamma, beta, eps):
    """x [B,T,C]: statistics over all B*T rows incl. padding."""
    flat = x.reshape(-1, x.shape[-1])
    mean = flat.mean(0)
    var = ((flat - mean) ** 2).mean(0)
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


def lstm_step(x, c, h, W, b):
    """tf.nn.rnn_cell.LSTMCell, gate order i,j,f,o, forget_bias 1.0. vectors."""
    z = np.concatenate([x, h]) @ W + b
    n = c.shape[0]
    i, j, f, o = z[:n], z[n:2 * n], z[2 * n:3 * n], z[3 * n:]
    c_new = sigmoid(f + 1.0) * c + sigmoid(i) * np.tanh(j)
    h_new = sigmoid(o) * np.tanh(c_new)
    return c_new, h_new


def zone(new, old, rate, training, seed, stream, b, T, t):
    """ZoneoutLSTMCell state update; idx = (b*T+t)*H + j."""
    H = new.shape[0]
    if training:
        if rate <= 0:
            return new
        idx = (np.uint64(b) * np.uint64(T) + np.uint64(t)) * np.uint64(H) + np.arange(H, dtype=np.uint64)
        return np.where(keep(seed, stream, idx, rate), new, old)
    return (1 - rate) * new + rate * old


def lstm_dir(x, length, W, b, H, reverse, zc, zh, training, seed, streams, bidx, T):
    """one sample, one direction of bidirectional_dynamic_rnn with sequence_length (module.py:93-108)."""
    out = np.zeros((T, H))
    c = np.zeros(H); h = np.zeros(H)
    steps = range(length - 1, -1, -1) if reverse else range(length)
    for t in steps:
        cn, hn = lstm_step(x[t], c, h, W, b)
        out[t] = hn
        c = zone(cn, c, zc, training, seed, streams[0], bidx, T, t)
        h = zone(hn, h, zh, training, seed, streams[1], bidx, T, t)
    return out


def mha_sample(x, P, prefix, heads, causal, rate, training, seed, stream, bidx):
    """MultiHeadAttention + SDPA for one sample (modules/self_attention.py:45-65,108-128). x [T,D]."""
    T, D = x.shape
    hd = D // heads
    kvq = dense(x, P[prefix + ".kvq.W"], P[prefix + ".kvq.b"])
    K, V, Q = kvq[:, :D], kvq[:, D:2 * D], kvq[:, 2 * D:]
    o = np.zeros((T, D))
    aligns = []
    for hh in range(heads):
        sl = slice(hh * hd, (hh + 1) * hd)
        s = Q[:, sl] @ K[:, sl].T / math.sqrt(hd)
        if causal:
            s = np.where(np.tril(np.ones((T, T), dtype=bool)), s, -np.inf)
        s = s - s.max(axis=1, keepdims=True)
        p = np.exp(s); p /= p.sum(axis=1, keepdims=True)
        aligns.append(p)
        pd = p
        if training and rate > 0:
            base = (np.uint64(bidx) * np.uint64(heads) + np.uint64(hh)) * np.uint64(T)
            idx = (base + np.arange(T, dtype=np.uint64)[:, None]) * np.uint64(T) + np.arange(T, dtype=np.uint64)
            pd = np.where(keep(seed, stream, idx, rate), p / (1 - rate), 0.0)
        o[:, sl] = pd @ V[:, sl]
    o = dense(o, P[prefix + ".o.W"], P[prefix + ".o.b"])
    tr = np.tanh(dense(o, P[prefix + ".t.W"], P[prefix + ".t.b"]))
    return x + tr, aligns


def encoder(batch, P, cfg, training, seed):
    src, slen = batch["source"], batch["source_length"]
    B, Ti = src.shape
    pre = np.zeros((B, Ti, cfg.enc_prenet[-1]))
    for b in range(B):
        x = P["embedding"][src[b]]
        for n in range(len(cfg.enc_prenet)):
            x = np.maximum(dense(x, P[f"enc.prenet{n}.W"], P[f"enc.prenet{n}.b"]), 0)
            x = dropout_rows(x, cfg.enc_prenet_drop, training, seed,
                             (rng.STREAM_ENC_PRENET0, rng.STREAM_ENC_PRENET1)[n], row0=b * Ti)
        pre[b] = x
    K = cfg.max_filter_width
    bank = np.concatenate([np.stack([conv1d_same_sample(pre[b], P[f"enc.bank{k}.W"]) for b in range(B)])
                           for k in range(1, K + 1)], axis=-1)
    bank = np.maximum(bn_train(bank, P["enc.bank.gamma"], P["enc.bank.beta"], cfg.bn_eps), 0)
    mp = bank.copy()
    mp[:, :-1] = np.maximum(bank[:, :-1], bank[:, 1:])
    p1 = np.stack([conv1d_same_sample(mp[b], P["enc.proj1.W"]) for b in range(B)])
    p1 = np.maximum(bn_train(p1, P["enc.proj1.gamma"], P["enc.proj1.beta"], cfg.bn_eps), 0)
    p2 = np.stack([conv1d_same_sample(p1[b], P["enc.proj2.W"]) for b in range(B)])
    p2 = bn_train(p2, P["enc.proj2.gamma"], P["enc.proj2.beta"], cfg.bn_eps)
    hw = p2 + pre
    H = cfg.cbhg_out_units // 2
    for n in range(cfg.num_highway):
        z = dense(hw, P[f"enc.highway{n}.W"], P[f"enc.highway{n}.b"])
        hh, tt = np.maximum(z[..., :H], 0), sigmoid(z[..., H:])
        hw = hh * tt + hw * (1 - tt)
    lstm_out = np.zeros((B, Ti, 2 * H))
    for b in range(B):
        lstm_out[b, :, :H] = lstm_dir(hw[b], int(slen[b]), P["enc.lstm_fw.W"], P["enc.lstm_fw.b"], H, False,
                                      cfg.zc, cfg.zh, training, seed,
                                      (rng.STREAM_ENC_LSTM_FW_C, rng.STREAM_ENC_LSTM_FW_H), b, Ti)
        lstm_out[b, :, H:] = lstm_dir(hw[b], int(slen[b]), P["enc.lstm_bw.W"], P["enc.lstm_bw.b"], H, True,
                                      cfg.zc, cfg.zh, training, seed,
                                      (rng.STREAM_ENC_LSTM_BW_C, rng.STREAM_ENC_LSTM_BW_H), b, Ti)
    if cfg.sa_units == 0:            # ZoneoutEncoderV1 (modules/module.py:336-339): pre-nets + CBHG, no self-attention branch
        return lstm_out, None, None
    sa_in = dense(lstm_out, P["enc.sa_proj.W"], P["enc.sa_proj.b"])
    # self_attention_num_hop stacked blocks, own weights each (modules/module.py:411-419, :433-439); alignments of the first hop
    x, enc_align = sa_in, None
    for hop in range(getattr(cfg, "sa_num_hop", 1)):
        sa_out = np.zeros_like(x)
        aligns = []
        for b in range(B):
            sa_out[b], al = mha_sample(x[b], P, hop_prefix("enc.sa", hop), cfg.sa_heads, False, cfg.sa_drop, training, seed,
                                       rng.STREAM_ENC_SA + HOP_STREAM * hop, b)
            aligns.append(al)
        if hop == 0:
            enc_align = [np.stack([aligns[b][h] for b in range(B)]) for h in range(cfg.sa_heads)]
        x = sa_out
    return lstm_out, sa_out, enc_align


def softmax_masked(e, length):
    out = np.zeros_like(e)
    v = e[:length]
    v = np.exp(v - v.max())
    out[:length] = v / v.sum()
    return out


def decoder(batch, lstm_out, sa_out, P, cfg, training, seed):
    slen = batch["source_length"]
    target = batch["mel"]
    B, Tm, nm = target.shape
    r = cfg.r; Td = Tm // r
    Ti = lstm_out.shape[1]
    feed = nm * cfg.n_feed_frame
    A, D = cfg.att_rnn_units, cfg.dec_units
    dec_out = np.zeros((B, Td, D))
    al1 = np.zeros((B, Td, Ti)); al2 = np.zeros((B, Td, Ti))
    k = cfg.att_kernel; pl = (k - 1) // 2
    Fk = P["dec.att1.F"][:, 0, :]
    for b in range(B):
        L = int(slen[b])
        tg = target[b].reshape(Td, nm * r)
        dec_in = np.concatenate([np.zeros((1, feed)), tg[:-1, -feed:]], axis=0)
        x = dec_in
        for n in range(len(cfg.dec_prenet)):
            y = dense(x, P[f"dec.prenet{n}.W"], P[f"dec.prenet{n}.b"])
            if n == 0 and cfg.num_speakers > 0:
                s = P["speaker_embedding"][int(batch["speaker_id"][b]) - cfg.speaker_offset]
                s = dense(s, P["dec.prenet0.Ws"], P["dec.prenet0.bs"])
                s = s / (1 + np.abs(s))
                y = np.maximum(y, 0) + s
                y = np.maximum(dense(y, P["dec.prenet0.W2"], P["dec.prenet0.b2"]), 0)
            else:
                y = np.maximum(y, 0)
            # apply_dropout_on_inference: the plain PreNet layers keep their dropout outside training (modules/module.py:564-577)
            on = training or (getattr(cfg, "apply_dropout_on_inference", False) and not (n == 0 and cfg.num_speakers > 0))
            x = dropout_rows(y, cfg.dec_prenet_drop, on, seed,
                             (rng.STREAM_DEC_PRENET0, rng.STREAM_DEC_PRENET1)[n], row0=b * Td)
        pre = x
        msk = (np.arange(Ti) < L)[:, None]
        v1 = lstm_out[b] * msk; k1 = v1 @ P["dec.att1.Wm"]
        dual = cfg.sa_units > 0       # one mechanism only in the baseline decoder (modules/module.py:566-574)
        if dual:
            v2 = sa_out[b] * msk; k2 = v2 @ P["dec.att2.Wm"]
        c0 = np.zeros(A); h0 = np.zeros(A); c1 = np.zeros(D); h1 = np.zeros(D); c2 = np.zeros(D); h2 = np.zeros(D)
        attn = np.zeros(cfg.ctx_dim)
        a_prev = np.zeros(Ti); alpha_prev = np.zeros(Ti); alpha_prev[0] = 1.0; u = 0.5
        agent = getattr(cfg, "transition_agent", False)
        for t in range(Td):
            cn, hn = lstm_step(np.concatenate([pre[t], attn]), c0, h0, P["dec.att_lstm.W"], P["dec.att_lstm.b"])
            c0 = zone(cn, c0, cfg.zc, training, seed, rng.STREAM_ATT_LSTM_C, b, Td, t)
            h0 = zone(hn, h0, cfg.zh, training, seed, rng.STREAM_ATT_LSTM_H, b, Td, t)
            # forward attention (modules/forward_attention.py:88-122)
            pq = hn @ P["dec.att.Wq"][:, :cfg.att1_units]
            f = np.zeros((Ti, cfg.att_filters))
            for tt in range(Ti):
                for j in range(k):
                    s = tt + j - pl
                    if 0 <= s < Ti:
                        f[tt] += a_prev[s] * Fk[j]
            f += P["dec.att1.bF"]
            lf = f @ P["dec.att1.U"]
            e = np.tanh(k1 + pq + lf + P["dec.att1.b"]) @ P["dec.att1.v"]
            a = softmax_masked(e, L)
            if getattr(cfg, "attention", "forward") == "location_sensitive":      # plain softmax alignments, no recursion
                al = a
            else:
                shifted = np.concatenate([[0.0], alpha_prev[:-1]])
                al = ((1 - u) * alpha_prev + u * shifted + 1e-7) * a
                al = al / al.sum()
                if agent:     # forward_attention.py:111-114: u of the NEXT step = sigmoid(Dense([context | processed query]))
                    zin = np.concatenate([al @ v1, pq])
                    u = float(sigmoid(zin @ P["dec.att1.Wa"][:, 0] + P["dec.att1.ba"][0]))
            a_prev = a + a_prev if getattr(cfg, "cumulative_weights", False) else a   # forward_attention.py:118-121
            alpha_prev = al
            # additive attention (BahdanauAttention; A.8)
            if dual:
                e2 = np.tanh(k2 + hn @ P["dec.att.Wq"][:, cfg.att1_units:]) @ P["dec.att2.v"]
                a2 = softmax_masked(e2, L)
                attn = np.concatenate([al @ v1, a2 @ v2])
            else:
                a2 = np.zeros(Ti)
                attn = al @ v1
            x1 = np.concatenate([hn, attn])
            cn1, hn1 = lstm_step(x1, c1, h1, P["dec.lstm1.W"], P["dec.lstm1.b"])
            c1 = zone(cn1, c1, cfg.zc, training, seed, rng.STREAM_LSTM1_C, b, Td, t)
            h1 = zone(hn1, h1, cfg.zh, training, seed, rng.STREAM_LSTM1_H, b, Td, t)
            cn2, hn2 = lstm_step(hn1, c2, h2, P["dec.lstm2.W"], P["dec.lstm2.b"])
            c2 = zone(cn2, c2, cfg.zc, training, seed, rng.STREAM_LSTM2_C, b, Td, t)
            h2 = zone(hn2, h2, cfg.zh, training, seed, rng.STREAM_LSTM2_H, b, Td, t)
            dec_out[b, t] = hn2; al1[b, t] = al; al2[b, t] = a2
    tr = dec_out                                                          # OutputAndStopTokenWrapper on the RNN output
    for hop in range(getattr(cfg, "dec_sa_num_hop", 1) if cfg.dec_sa_units > 0 else 0):      # modules/module.py:707-715, :753-757
        nxt = np.zeros_like(tr)
        for b in range(B):
            nxt[b], _ = mha_sample(tr[b], P, hop_prefix("dec.sa", hop), cfg.dec_sa_heads, True, cfg.dec_sa_drop, training, seed,
                                   rng.STREAM_DEC_SA + HOP_STREAM * hop, b)
        tr = nxt
    y = dense(tr, P["dec.out.W"], P["dec.out.b"])
    mel = y[..., :-1].reshape(B, Tm, nm)
    stop = y[..., -1:]
    return mel, stop, al1, al2, dec_out


def forward(P, batch, cfg, training=True, seed=0):
    """model_fn forward + losses (models/models.py:278-482)."""
    P = {k: np.asarray(v, dtype=np.float64) for k, v in P.items()}
    lstm_out, sa_out, enc_align = encoder(batch, P, cfg, training, seed)
    mel, stop, al1, al2, dec_out = decoder(batch, lstm_out, sa_out, P, cfg, training, seed)
    w = batch["spec_loss_mask"][:, :, None]
    mel_loss = (np.abs(mel - batch["mel"]) * w).sum() / (mel.shape[-1] * batch["spec_loss_mask"].sum())
    x = stop[..., 0]; z = batch["done"]
    bce = np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))
    done_loss = (bce * batch["binary_loss_mask"]).sum() / batch["binary_loss_mask"].sum()
    return dict(mel=mel, stop=stop, alignment1=al1, alignment2=al2, enc_alignment=enc_align, lstm_out=lstm_out,
                sa_out=sa_out, dec_out=dec_out, mel_loss=mel_loss, done_loss=done_loss, loss=mel_loss + done_loss)
