"""Step-by-step decode with is_training=False: free-running synthesis and the validation (teacher-fed) pass.

Replaces the inference branch of the reference: RNNTransformer else-branch (modules/module.py:762-778) =
RNNStateHistoryWrapper (modules/rnn_wrappers.py:47-80) + TransformerWrapper (:87-124) +
OutputAndStopTokenTransparentWrapper (:188-214) driven by StopTokenBasedInferenceHelper / ValidationHelper
(mirrors in modules/helpers.py:58-166); SURVEY.md A.14, config 5 of BASELINE.json.

The reference re-runs the causal self-attention over the WHOLE decoder-output history at every step (O(T^2) per
step).  Because the mask is causal and there is no padding mask, that is mathematically identical to attending with
the new query row over cached keys/values, which is what happens here: the K|V|Q projection of step t is written
into row t of a [B, Tmax, 3D] buffer (the KV cache), the score row, softmax row and P.V row are three tiny launches.
The recurrent part reuses the training kernels unchanged: the cluster kernels process the time range [t, t+1) and
restart from the tensors they saved at step t-1, exactly like a pipeline chunk.  Zoneout runs in interpolation
mode, dropout is off, BatchNorm uses the moving statistics.
"""
import math

import torch

from . import ops
from ._lib import SattError
from .ops import ACT_NONE, ACT_RELU, ACT_SOFTSIGN, ACT_TANH
from .engine import S_ATT_C, S_ATT_H, S_L1_C, S_L1_H, S_L2_C, S_L2_H


def infer(eng, source, source_length, max_steps=None, teacher=None, speaker_id=None, min_steps=10, stop_threshold=0.5,
          check_every=1, teacher_alignments=None):
    """eng: Engine.  source int64 [B,Ti], source_length int64 [B] (device tensors or array-likes).
    teacher=None: free running, at most max_steps decoder steps, stops when sigmoid(stop) > stop_threshold for every
    sample and t > min_steps (checked every `check_every` steps: one host sync each).
    teacher=[B,Tm,num_mels]: inputs from the ground truth (validation pass), exactly Tm/r steps.
    teacher_alignments=(a1, a2), each [B,T,Ti] with T >= the number of steps: forced-alignment mode
    (use_forced_alignment_mode: modules/teacher_forcing_attention.py:13-78, models/models.py:411-428) - both attention
    mechanisms return the given alignment of the step; contexts and alignment histories follow them.
    Returns dict(mel [B,T*r,num_mels], stop [B,T,1], alignment1 [B,T,Ti], alignment2 [B,T,Ti], steps=T,
    lstm_out, sa_out)."""
    c, P, dev = eng.cfg, eng.P, eng.dev
    f32 = dict(dtype=torch.float32, device=dev)
    batch = {"source": torch.as_tensor(source).to(dev).contiguous(),
             "source_length": torch.as_tensor(source_length).to(dev).contiguous()}
    if c.num_speakers > 0:
        batch["speaker_id"] = torch.as_tensor(speaker_id).to(dev).contiguous()
    B, Ti = batch["source"].shape
    slen = batch["source_length"]
    nm, r = c.num_mels, c.r
    feed = nm * c.n_feed_frame
    if teacher is not None:
        teacher = torch.as_tensor(teacher, **f32).contiguous()
        Td = teacher.shape[1] // r
        tg = teacher.view(B, Td, nm * r)
    else:
        if not max_steps or max_steps < 1:
            raise SattError("infer: max_steps must be given for free-running decode")
        Td = int(max_steps)
    ta1 = ta2 = None
    if teacher_alignments is not None:
        ta1 = torch.as_tensor(teacher_alignments[0], **f32).contiguous()
        ta2 = torch.as_tensor(teacher_alignments[1] if teacher_alignments[1] is not None else teacher_alignments[0],
                              **f32).contiguous()       # single source: the second history is ignored
        if ta1.shape != ta2.shape or ta1.shape[0] != B or ta1.shape[2] != Ti or ta1.shape[1] < Td:
            raise SattError("infer: teacher_alignments must be two [B, T >= steps, Ti] tensors")
        if ta1.shape[1] != Td:                       # the kernels index rows as (b * Td + t)
            ta1 = ta1[:, :Td].contiguous(); ta2 = ta2[:, :Td].contiguous()
    ctx = {"training": False, "batch": batch}
    lstm_out, sa_out = eng._encode(batch, False, ctx)
    M, Md = B * Ti, B * Td
    E = lambda *s: torch.empty(*s, **f32)
    Z = lambda *s: torch.zeros(*s, **f32)

    # ---- memories, attention parameters and per-step buffers (same layouts as Engine.forward)
    V1, V2, U1, U2, A, D = c.cbhg_out_units, c.sa_units, c.att1_units, c.att2_units, c.att_rnn_units, c.dec_units
    CT, G4 = V1 + V2, 4 * A
    values1, keys1 = E(M, V1), E(M, U1)
    ops.seq_mask(lstm_out, slen, values1, B, Ti, V1)
    ops.linear(values1, P["dec.att1.Wm"], None, keys1)
    values2 = keys2 = None            # single attention source (ExtendedDecoder): NULL second mechanism, see Engine.forward
    if c.dual:
        values2, keys2 = E(M, V2), E(M, U2)
        ops.seq_mask(sa_out, slen, values2, B, Ti, V2)
        ops.linear(values2, P["dec.att2.Wm"], None, keys2)
    pn = c.dec_prenet[-1]
    xg_att, att_out = Z(Md, G4), Z(Md, A + CT)
    al1, al2, a1 = Z(B, Td, Ti), Z(B, Td, Ti), Z(B, Td, Ti)
    pq, flb = Z(Md, U1 + U2), Z(Md * Ti, c.att_filters)
    ag, acn, acs, ahs = Z(Md, G4), Z(Md, A), Z(Md, A), Z(Md, A)
    ap = ops.attn_rnn_params(
        B=B, Td=Td, Ti=Ti, A=A, U1=U1, V1=V1, U2=U2, V2=V2, kernel=c.att_kernel, filters=c.att_filters, training=0,
        keys_lds_bf16=int(ops.get_precision() == "bf16"), zc=c.zc, zh=c.zh, zc_thresh=0, zh_thresh=0, seed=eng.seed,
        stream_c=S_ATT_C, stream_h=S_ATT_H, lengths=slen, xg=xg_att, Wrec=eng.shadow["att.Wrec"],
        Wq=eng.shadow["att.Wq"], keys1=keys1, values1=values1, keys2=keys2, values2=values2,
        locF=P["dec.att1.F"], locFb=P["dec.att1.bF"], locU=P["dec.att1.U"], v1=P["dec.att1.v"],
        b1=P["dec.att1.b"], v2=P.get("dec.att2.v"), out=att_out, align1=al1, align2=al2, a1=a1, pq=pq,
        fl=flb, gates=ag, cnew=acn, cstate=acs, hstate=ahs, teach1=ta1, teach2=ta2,
        att1_mode=int(c.attention == "location_sensitive"), cumulative=int(c.cumulative_weights),
        acum=Z(B, Td, Ti) if c.cumulative_weights else None)
    Ca = ops.attn_cluster_size(ap)
    Cn = ops.lstm_cluster_size(B, D)
    if not Ca or not Cn:
        raise SattError("infer: this shape is not supported by the cluster kernels (the incremental decode restarts "
                        "them step by step)")
    if Ca not in eng._pack_cache:
        eng._pack_cache[Ca] = ops.attn_cluster_pack(P["dec.att_lstm.W"][pn:], A, Ca)
    aws = ops.attn_cluster_ws(ap, Ca, dev)
    lp1, lp2 = eng.lstm_cluster_packs(Cn)
    cws1, cws2 = ops.lstm_cluster_ws(B, D, Cn, dev), ops.lstm_cluster_ws(B, D, Cn, dev)
    xg1, xg2 = Z(1, Md, 4 * D), Z(1, Md, 4 * D)
    h1, dec_out = Z(Md, D), Z(Md, D)
    l1 = (Z(1, Md, 4 * D), Z(1, Md, D), Z(1, Md, D), Z(1, Md, D))
    l2 = (Z(1, Md, 4 * D), Z(1, Md, D), Z(1, Md, D), Z(1, Md, D))
    Ds, heads = c.dec_sa_units, c.dec_sa_heads
    hd = Ds // heads
    kvq = Z(Md, 3 * Ds) if Ds else None     # the KV cache: rows (b, t) = K | V | Q of step t
    NO = nm * r + 1
    yout = Z(Md, NO)
    step_view = lambda buf, t: buf.view(B, Td, -1)[:, t]          # [B, C] rows (b, t), leading dimension Td*C

    # multi-speaker pre-net term (constant over time): softsign(emb[speaker] Ws + bs)
    sproj = None
    if c.num_speakers > 0:
        semb = E(B, c.speaker_dim)
        ops.embedding_fwd(batch["speaker_id"], P["speaker_embedding"], semb, offset=c.speaker_offset)
        sproj = E(B, c.dec_prenet[0])
        ops.linear(semb, P["dec.prenet0.Ws"], P["dec.prenet0.bs"], sproj, act=ACT_SOFTSIGN)

    x_in = Z(B, feed)                                              # go frame
    s_row, p_row = E(B * heads, Td), E(B * heads, Td)
    o_t, o2_t, th_t, tr_t = E(B, Ds), E(B, Ds), E(B, Ds), E(B, Ds)
    steps = 0
    for t in range(Td):
        # ---- pre-net of the fed-back frame (dropout off)
        x = x_in
        for n, o in enumerate(c.dec_prenet):
            y = E(B, o)
            if n == 0 and sproj is not None:
                d0 = E(B, o)
                ops.linear(x, P["dec.prenet0.W"], P["dec.prenet0.b"], d0, act=ACT_RELU)
                ops.axpby(sproj, d0, 1.0, 1.0)
                ops.linear(d0, P["dec.prenet0.W2"], P["dec.prenet0.b2"], y, act=ACT_RELU)
            else:
                ops.linear(x, P[f"dec.prenet{n}.W"], P[f"dec.prenet{n}.b"], y, act=ACT_RELU)
            x = y
        # ---- attention RNN, LSTM1, LSTM2: the training kernels on the time range [t, t+1)
        ops.linear(x, P["dec.att_lstm.W"][:pn], P["dec.att_lstm.b"], step_view(xg_att, t))
        ops.attn_cluster_fwd(ap, Ca, eng._pack_cache[Ca][0], aws, t, t + 1)
        ops.linear(step_view(att_out, t), P["dec.lstm1.W"][:A + CT], P["dec.lstm1.b"], step_view(xg1[0], t))
        ops.lstm_cluster_fwd(xg1, lp1[0], B, Td, D, Cn, False, c.zc, c.zh, eng.seed, S_L1_C, S_L1_H, h1,
                             *l1, cws1, t, t + 1)
        ops.linear(step_view(h1, t), P["dec.lstm2.W"][:D], P["dec.lstm2.b"], step_view(xg2[0], t))
        ops.lstm_cluster_fwd(xg2, lp2[0], B, Td, D, Cn, False, c.zc, c.zh, eng.seed, S_L2_C, S_L2_H, dec_out,
                             *l2, cws2, t, t + 1)
        # ---- causal self-attention of the new row over the KV cache (== re-running it over the whole history)
        xt = step_view(dec_out, t)
        yt = step_view(yout, t)
        if not Ds:          # ExtendedDecoder: the projections read the DecoderRNNV2 output (OutputAndStopTokenWrapper)
            ops.linear(xt, P["dec.out.W"], P["dec.out.b"], yt)
        else:
          ops.linear(xt, P["dec.sa.kvq.W"], P["dec.sa.kvq.b"], step_view(kvq, t))
          n = t + 1
          ops.gemm(1, n, hd, kvq[t:, 2 * Ds:], 3 * Ds, kvq, 1, 3 * Ds, s_row, Td, batch=(B, heads),
                   sA=(Td * 3 * Ds, hd), sB=(Td * 3 * Ds, hd), sC=(heads * Td, Td))
          ops.softmax_rows(s_row, p_row, B * heads, n, 1.0 / math.sqrt(hd))
          ops.gemm(1, hd, n, p_row, Td, kvq[:, Ds:], 3 * Ds, 1, o_t, Ds, batch=(B, heads),
                   sA=(heads * Td, Td), sB=(Td * 3 * Ds, hd), sC=(Ds, hd))
          ops.linear(o_t, P["dec.sa.o.W"], P["dec.sa.o.b"], o2_t)
          ops.linear(o2_t, P["dec.sa.t.W"], P["dec.sa.t.b"], th_t, act=ACT_TANH)
          ops.axpby(xt, tr_t, 1.0, 0.0)
          ops.axpby(th_t, tr_t, 1.0, 1.0)
          ops.linear(tr_t, P["dec.out.W"], P["dec.out.b"], yt)
        steps = t + 1
        # ---- next input / stop rule (modules/helpers.py:94,103-107,157-158 mirrors)
        if teacher is not None:
            x_in = tg[:, t, nm * r - feed:]
        else:
            x_in = yt[:, nm * r - feed:nm * r]
            if t > min_steps and (t % check_every == 0 or t == Td - 1):
                if bool((torch.sigmoid(yt[:, NO - 1]) > stop_threshold).all()):
                    break
    ops.attn_cluster_status(ap, Ca, aws)
    ops.lstm_cluster_status(cws1, B, D, Cn)
    ops.lstm_cluster_status(cws2, B, D, Cn)
    y = yout.view(B, Td, NO)[:, :steps]
    return dict(yout=yout, mel=y[:, :, :NO - 1].reshape(B, steps * r, nm), stop=y[:, :, NO - 1:].contiguous(),
                alignment1=al1[:, :steps], alignment2=al2[:, :steps], steps=steps,
                lstm_out=lstm_out.view(B, Ti, -1), sa_out=sa_out.view(B, Ti, -1) if c.dual else None,
                enc_alignment=ctx["enc_align"].view(B, c.sa_heads, Ti, Ti) if c.dual else None)


def postnet_infer(eng, mel):
    """PostNetV2 in PREDICT / EVAL mode (reference models/models.py:440-462): num_layers x [Conv1d(k) ->
    BatchNorm with the MOVING statistics -> tanh (last layer: linear)], dropout off, Dense(C -> num_mels), residual.
    mel [B, T, num_mels] (device) -> mel_postnet [B, T, num_mels]."""
    c, P = eng.cfg, eng.P
    if not c.use_postnet_v2:
        raise SattError("postnet_infer: the model was built without use_postnet_v2")
    mel = mel.contiguous()
    B, T, nm = mel.shape
    x = mel.view(B * T, nm)
    L, Co = c.num_postnet_v2_layers, c.postnet_v2_out_channels
    for n in range(L):
        pre = torch.empty(B * T, Co, dtype=torch.float32, device=eng.dev)
        ops.conv1d(x, T, P[f"postnet.conv{n}.W"], pre)
        y = torch.empty_like(pre)
        ops.bn_infer(pre, P[f"postnet.bn{n}.gamma"], P[f"postnet.bn{n}.beta"], eng.bn[f"postnet{n}"][0],
                     eng.bn[f"postnet{n}"][1], y, c.bn_eps, ACT_TANH if n < L - 1 else ACT_NONE)
        x = y
    out = mel.view(B * T, nm).clone()
    ops.linear(x, P["postnet.proj.W"], P["postnet.proj.b"], out, residual=out)
    return out.view(B, T, nm)


def evaluate(eng, batch, speaker_id=None):
    """EVAL double pass of the reference's model_fn (models/models.py:517-564): (1) the free-running decode over
    exactly Td = Tm/r steps (ValidationHelper with teacher_forcing=False: own outputs fed back, no stop rule) and
    (2) the teacher-fed validation pass, each scored with the training losses (spec_loss + binary_loss with the
    batch's masks).  Returns the scalars under the reference's metric names plus the free run's outputs."""
    b = eng.to_device_batch({k: v for k, v in batch.items() if hasattr(v, "dtype") or isinstance(v, torch.Tensor)})
    c = eng.cfg
    B, Tm = b["mel"].shape[0], b["mel"].shape[1]
    Td = Tm // c.r
    nm = c.num_mels
    NO = nm * c.r + 1
    spk = b.get("speaker_id") if speaker_id is None else speaker_id

    def score(out):
        y = out["yout"]
        dy = torch.empty_like(y)
        ls = torch.zeros(3, dtype=torch.float32, device=eng.dev)
        ops.loss_fwd_bwd(y, NO, b["mel"], b["spec_loss_mask"], y[:, NO - 1:], NO, b["done"], b["binary_loss_mask"],
                         B, Tm, nm, Td, eng.loss_l2, ls, dy, NO, dy[:, NO - 1:], NO, eng._loss_ws)
        return [float(x) for x in ls.cpu()]
    free = infer(eng, b["source"], b["source_length"], max_steps=Td, min_steps=1 << 30, speaker_id=spk)
    mel_loss, done_loss, loss = score(free)
    tf = infer(eng, b["source"], b["source_length"], teacher=b["mel"], speaker_id=spk)
    mel_t, done_t, loss_t = score(tf)
    return dict(mel_loss=mel_loss, done_loss=done_loss, loss=loss, mel_loss_with_teacher=mel_t,
                done_loss_with_teacher=done_t, loss_with_teacher=loss_t, mel=free["mel"], stop=free["stop"],
                alignment1=free["alignment1"], alignment2=free["alignment2"], mel_with_teacher=tf["mel"])
