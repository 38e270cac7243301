"""Step-by-step decode with is_training=False: free-running synthesis and the validation (teacher-fed) pass.

Replaces the inference branch of the reference: RNNTransformer else-branch (modules/module.py:762-778) =
RNNStateHistoryWrapper (modules/rnn_wrappers.py:47-80) + TransformerWrapper (:87-124) +
OutputAndStopTokenTransparentWrapper (:188-214) driven by StopTokenBasedInferenceHelper / ValidationHelper
(mirrors in modules/helpers.py:58-166); SURVEY.md A.14, config 5 of BASELINE.json.

The reference re-runs the causal self-attention over the WHOLE decoder-output history at every step (O(T^2) per
step).  Because the mask is causal and there is no padding mask, that is mathematically identical to attending with
the new query row over cached keys/values, which is what happens here: the K|V|Q projection of step t is written
into row t of a [B, Tmax, 3D] buffer (the KV cache) and one kernel computes the score row, its softmax and P.V.

One decoder step is ~17 small launches of csrc/decode.hip whose time index lives in device memory, so the step is
captured ONCE as a hipGraph (several steps per graph) and replayed: the host issues one graph launch per
`steps_per_graph` steps and reads the device-side stop flag one replay behind.  Zoneout runs in interpolation mode,
dropout is off, BatchNorm uses the moving statistics.
"""
import math

import torch

from . import ops
from ._lib import SattError
from .ops import ACT_NONE, ACT_RELU, ACT_SOFTSIGN, ACT_TANH


class DecodeSession:
    """Buffers, kernel parameter blocks and the captured hipGraph of the decoder step for one problem shape
    (B, Ti, Td, feeding mode, forced alignments, stop rule).  Kept on the engine and reused by later calls of the same
    shape: a new utterance only refills the memories and resets the recurrent state."""

    def __init__(self, eng, B, Ti, Td, teacher, forced, min_steps, stop_threshold, steps_per_graph, use_graph):
        c, P, dev = eng.cfg, eng.P, eng.dev
        self.eng, self.B, self.Ti, self.K = eng, B, Ti, max(1, int(steps_per_graph))
        self.Td = Td
        # The persistent kernel (csrc/decode_mega2.hip) is decided FIRST, by the library itself (satt_dec_mega_supported on the
        # shape block - one condition, one place), because the launch granularity follows from it: the kernel leaves its step loop
        # at the stop token by itself, so one launch may span many steps - the launch prologue (weights into registers, tables
        # into LDS: ~10 us) is spread over MEGA_STEPS steps and nothing runs past the stop token but one launch that returns at
        # once.  The hipGraph path keeps `steps_per_graph` (the host polls the stop flag once per replay).
        V1, V2, U1, U2, A, D, Ds = c.cbhg_out_units, c.sa_units, c.att1_units, c.att2_units, c.att_rnn_units, c.dec_units, c.dec_sa_units
        nm, r = c.num_mels, c.r
        feed, NO = nm * c.n_feed_frame, nm * r + 1
        self._mega_shape = None
        if (self.MEGA and use_graph and not forced and c.dual and c.num_speakers == 0 and not c.transition_agent and
                not c.apply_dropout_on_inference and len(c.dec_prenet) == 2 and Ds and c.dec_sa_num_hop == 1 and
                ops.get_precision() == "bf16" and B <= self.MEGA_MAX_B):
            shape = dict(B=B, Td=Td, Ti=Ti, A=A, D=D, Ds=Ds, heads=c.dec_sa_heads, U1=U1, V1=V1, U2=U2, V2=V2, kernel=c.att_kernel,
                         filters=c.att_filters, att1_mode=int(c.attention == "location_sensitive"), cumulative=int(c.cumulative_weights),
                         P0=c.dec_prenet[0], P1=c.dec_prenet[1], feed=feed, NO=NO, ldout=(NO + 7) // 8 * 8, zc=c.zc, zh=c.zh,
                         stop_threshold=float(stop_threshold), min_steps=int(min_steps))
            if ops.dec_mega_supported(ops.dec_mega_params(**shape)):
                self._mega_shape = shape
                self.K = max(self.K, self.MEGA_STEPS)
        Tdp = self.Tdp = (Td + self.K - 1) // self.K * self.K          # whole graphs: rows past Td are scratch
        f32 = dict(dtype=torch.float32, device=dev)
        Z = lambda *s: torch.zeros(*s, **f32)
        CT, UQ = V1 + V2, U1 + U2
        # Two copies of the step counter: sB is read by the pre-net launches (the first of a step) and written by the
        # output projection (the last); the last pre-net launch copies it into sA, which every other launch reads.  Each
        # word is only ever written by a launch none of whose workgroups reads it (csrc/decode.hip: step bookkeeping).
        self.steps2 = torch.zeros(2, dtype=torch.int32, device=dev)
        self.step, sB = self.steps2[0:1], self.steps2[1:2]
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lengths = torch.zeros(B, dtype=torch.int64, device=dev)
        self.values1, self.keys1 = Z(B * Ti, V1), Z(B * Ti, U1)
        self.values2 = Z(B * Ti, V2) if c.dual else None
        self.keys2 = Z(B * Ti, U2) if c.dual else None
        self.yout = Z(B, Tdp + 1, NO)                     # row 0 = go frame (zeros); step t writes row t + 1
        self.tin = Z(B, Tdp, feed) if teacher else None   # teacher-fed inputs: go frame | shifted targets
        self.sproj = Z(B, c.dec_prenet[0]) if c.num_speakers > 0 else None
        self.ctx = Z(2, B, CT)                            # contexts of step t in buffer t & 1 (csrc/decode.hip)
        self.a_state, self.alpha_state = Z(2, B, Ti), Z(2, B, Ti)      # double-buffered by step parity
        self.e1, self.e2 = Z(B, Ti), Z(B, Ti)
        self.al1, self.al2 = Z(B, Tdp, Ti), (Z(B, Tdp, Ti) if c.dual else None)
        self.teach1 = Z(B, Tdp, Ti) if forced else None
        self.teach2 = Z(B, Tdp, Ti) if (forced and c.dual) else None
        # c, h of the three cells, double-buffered by step parity (csrc/decode.hip: read [t & 1], write [(t & 1) ^ 1])
        self.states = [Z(2, B, A), Z(2, B, A), Z(2, B, D), Z(2, B, D), Z(2, B, D), Z(2, B, D)]
        ca, ha, c1, h1, c2, h2 = self.states
        hq, pq, h1n, dout = Z(B, A), Z(2, B, UQ), Z(B, D), Z(B, D)     # pq: processed query of step t in buffer t & 1
        NH = c.dec_sa_num_hop if Ds else 0                # stacked causal blocks (modules/module.py:707-715): one K|V|Q cache per hop
        self.kvqs = [Z(B, Tdp, 3 * Ds) for _ in range(NH)]
        self.kvq = self.kvqs[0] if NH else None
        o_t, tr_t = (Z(B, Ds), Z(B, Ds)) if Ds else (None, None)
        st = self.step
        self._keep = [hq, pq, h1n, dout, o_t, tr_t]
        L = []          # the step: a list of (launcher, parameter block) pairs
        # LSTM weights with the gate columns regrouped per block of 8 units (csrc/decode.hip, LSTM form), in the
        # precision of the run; refilled from the parameters by refresh_folded()
        wdt = torch.bfloat16 if ops.get_precision() == "bf16" else torch.float32
        self.lstm_w = {n: torch.empty(P[n].shape, dtype=wdt, device=dev) for n in ("dec.att_lstm.W", "dec.lstm1.W", "dec.lstm2.W")}
        # mel | stop projection with its rows padded to a multiple of 4 columns (zeros): 16-byte / 8-byte weight loads
        self.out_w = torch.zeros(P["dec.out.W"].shape[0], (NO + 7) // 8 * 8, dtype=wdt, device=dev)[:, :NO]

        def lin(xs, W, y, step=st, graph=True, **kw):
            prm = ops.dec_linear_params(xs, W, y, step=step, B=B, **kw)
            if graph:
                L.append((ops.dec_linear, prm))
        # stop logit of step t = last column of output row t + 1 (evaluated on the device, one step later)
        stop_rule = None if teacher else (self.yout.view(-1)[NO + NO - 1:], (Tdp + 1) * NO, NO, self.flag, stop_threshold, min_steps)
        # ---- pre-net of the fed-back frame (MultiSpeakerPreNet: modules/multi_speaker_modules.py:27-32).  Dropout is off unless
        #      apply_dropout_on_inference keeps it on in the plain PreNet layers (modules/module.py:564-577): mask of row
        #      (b, step) of a [B, Td, units] activation, seeded by the engine's device seed word
        from .engine import S_DEC_PRENET0, S_DEC_PRENET1
        # the session's OWN seed word (the captured kernels read it in place): infer() refreshes it per call - the reference
        # draws fresh masks per synthesis run, which is the point of the flag (output variation); eng.seed only advances
        # with optimiser steps
        self.drop_seed = eng.seed.clone()
        pdrop = lambda n: dict(drop=ops.Drop(c.dec_prenet_drop, (S_DEC_PRENET0, S_DEC_PRENET1)[n], self.drop_seed), drop_T=Td) \
            if c.apply_dropout_on_inference else {}
        if teacher:
            x = (self.tin, feed, Tdp * feed, feed)
        else:           # the last n_feed_frame frames of the previous step's output (modules/helpers.py:94,157-158 mirrors)
            x = (self.yout.view(-1)[nm * r - feed:], feed, (Tdp + 1) * NO, NO)
        for n, o in enumerate(c.dec_prenet):
            y = Z(B, o); self._keep.append(y)
            last = dict(step_out=(st, 0)) if n == len(c.dec_prenet) - 1 else {}
            first = dict(stop=stop_rule) if n == 0 else {}
            if n == 0 and self.sproj is not None:
                d0 = Z(B, o); self._keep.append(d0)
                lin([x], eng.W("dec.prenet0.W"), (d0, o, 0), step=sB, bias=P["dec.prenet0.b"], act=ACT_RELU,
                    res=(self.sproj, o, 0), **first)
                lin([(d0, o, o, 0)], eng.W("dec.prenet0.W2"), (y, o, 0), step=sB, bias=P["dec.prenet0.b2"], act=ACT_RELU, **last)
            else:
                lin([x], eng.W(f"dec.prenet{n}.W"), (y, o, 0), step=sB, bias=P[f"dec.prenet{n}.b"], act=ACT_RELU, **first, **last,
                    **pdrop(n))
            x = (y, o, o, 0)
        # ---- attention RNN cell: [pre-net | attention_{t-1} | h] (AttentionWrapper step)
        # attention_{t-1} = the context buffer of the OTHER parity: base at buffer 1, parity stride -B*CT
        lin([x, (self.ctx[1], CT, CT, 0, -B * CT), (ha, A, A, 0, B * A)], self.lstm_w["dec.att_lstm.W"], (hq, A, 0),
            bias=P["dec.att_lstm.b"], lstm=(A, ca, ha, c.zc, c.zh))
        wq = eng.W("dec.att.Wq")         # the query layer runs inside the attention kernel
        lin([(hq, A, A, 0)], wq, (pq[0], UQ, 0), graph=False)
        bf = ops.get_precision() == "bf16"
        self.att = ops.dec_attention_params(
            A=A, hq=hq, Wq=None if bf else wq.w, Wqb=wq.n if bf else None, pq_out=pq,
            agentW=P["dec.att1.Wa"] if c.transition_agent else None, agentb=P["dec.att1.ba"] if c.transition_agent else None,
            B=B, Td=Tdp, Ti=Ti, U1=U1, V1=V1, U2=U2, V2=V2, kernel=c.att_kernel, filters=c.att_filters,
            att1_mode=int(c.attention == "location_sensitive"), cumulative=int(c.cumulative_weights), lengths=self.lengths,
            keys1=self.keys1, values1=self.values1, keys2=self.keys2, values2=self.values2, locF=P["dec.att1.F"],
            locFb=P["dec.att1.bF"], locU=P["dec.att1.U"], v1=P["dec.att1.v"], b1=P["dec.att1.b"], v2=P.get("dec.att2.v"),
            teach1=self.teach1, teach2=self.teach2, a_state=self.a_state, alpha_state=self.alpha_state, e1=self.e1, e2=self.e2, ctx=self.ctx,
            align1=self.al1, align2=self.al2, step=st)
        L.append((ops.dec_attention, self.att))
        # ---- DecoderRNNV2: two ZoneoutLSTM cells on [h_att | attention_t]
        lin([(hq, A, A, 0), (self.ctx, CT, CT, 0, B * CT), (h1, D, D, 0, B * D)], self.lstm_w["dec.lstm1.W"], (h1n, D, 0),
            bias=P["dec.lstm1.b"], lstm=(D, c1, h1, c.zc, c.zh))
        lin([(h1n, D, D, 0), (h2, D, D, 0, B * D)], self.lstm_w["dec.lstm2.W"], (dout, D, 0), bias=P["dec.lstm2.b"],
            lstm=(D, c2, h2, c.zc, c.zh))
        yrow = (self.yout.view(-1)[NO:], (Tdp + 1) * NO, NO)
        if Ds:          # causal self-attention of the new row over the KV cache, then SelfAttentionTransformer's tail
            from .params import sa_prefix
            heads = c.dec_sa_heads
            # output projection and the transformer's Dense are both linear: tanh((o Wo + bo) Wt + bt) = tanh(o Wot + bot)
            # with Wot = Wo Wt, bot = bo Wt + bt folded per call (refresh_folded) - one launch instead of two
            self.Wot, self.bot, self.Wot_k = [], [], []
            xh = dout                   # input row of the hop: the DecoderRNNV2 output, then the previous hop's output row
            for h in range(NH):         # row t of hop h depends on rows <= t of hop h - 1 only (causal): the caches advance together
                pre, cache = sa_prefix("dec.sa", h), self.kvqs[h]
                lin([(xh, Ds, Ds, 0)], eng.W(pre + ".kvq.W"), (cache, Tdp * 3 * Ds, 3 * Ds), bias=P[pre + ".kvq.b"])
                oh = o_t if h == 0 else Z(B, Ds)
                L.append((lambda _, cache=cache, oh=oh: ops.dec_self_attn(cache, oh, st, B, Tdp, Ds, heads, 1.0 / math.sqrt(Ds // heads)), None))
                Wot, bot = Z(Ds, Ds), Z(1, Ds)
                Wk = torch.empty(Ds, Ds, dtype=wdt, device=dev) if wdt != torch.float32 else Wot       # in the run's precision
                self.Wot.append(Wot); self.bot.append(bot); self.Wot_k.append(Wk)
                yh = tr_t if h == NH - 1 else Z(B, Ds)
                self._keep += [oh, yh]
                lin([(oh, Ds, Ds, 0)], Wk, (yh, Ds, 0), bias=bot, act=ACT_TANH, res=(xh, Ds, 0))
                xh = yh
            lin([(tr_t, Ds, Ds, 0)], self.out_w, yrow, bias=P["dec.out.b"], step_out=(sB, 1))
        else:           # ExtendedDecoder: the projections read the DecoderRNNV2 output (OutputAndStopTokenWrapper)
            lin([(dout, D, D, 0)], self.out_w, yrow, bias=P["dec.out.b"], step_out=(sB, 1))
        self.launches = self._fuse_pairs(self._fuse_chains(L)) if self.FUSE else L
        # kernel launches per decoder step (the attention entry is two kernels unless the alignments are forced)
        self.kernel_launches = sum(2 if (fn is ops.dec_attention and not forced) else 1 for fn, _ in self.launches)
        self.graph = None
        # ---- persistent form (csrc/decode_mega2.hip): the same step, ONE launch per K steps on 32 persistent workgroups that exchange
        # {tag, value} granules instead of nine dependent launches - for the configurations satt_dec_mega_supported took above
        self.mega = None
        self.ctab = self._ctw = None
        if self._mega_shape is not None:
            from .params import sa_prefix
            pre = sa_prefix("dec.sa", 0)
            self._mega_err = torch.zeros(1, dtype=torch.int32, device=dev)          # sticky error word
            self._mega_part = Z(max(1, ops.dec_mega_scratch_floats(B, c.dec_sa_heads, Ds // c.dec_sa_heads)))
            # context tables values W_c (the cells take  sum_r alpha_r (values_r W_c)  instead of ctx W_c):
            # [B * Ti][LSTM 1 x values1 | LSTM 1 x values2 | attention LSTM x values1 | attention LSTM x values2][4 * 256]
            self.ctab = Z(B * Ti, 4 * 4 * D)
            self._ctw = [Z(v, 4 * D) for v in (V1, V2, V1, V2)]
            # folded feedback (free running): pre-net layer 0 straight from the output transform's result - fed = the last `feed` mel
            # columns of y = vc Wout + bout is linear in vc, so relu(fed Wp0 + bp0) = relu(vc Wf + bf); Wf as bf16 hi + lo (refresh_folded)
            P0w = c.dec_prenet[0]
            self._fb = None if (teacher or not self.MEGA_FOLD_FEEDBACK) else dict(
                a=Z(Ds, feed), b=Z(feed, P0w), w=Z(Ds, P0w), t=Z(Ds, P0w), bias=Z(1, P0w),
                hi=torch.zeros(Ds, P0w, dtype=torch.bfloat16, device=dev), lo=torch.zeros(Ds, P0w, dtype=torch.bfloat16, device=dev))
            self.mega = ops.dec_mega_params(
                **dict(self._mega_shape, Td=Tdp),
                Wp0=eng.W("dec.prenet0.W").n, Wp1=eng.W("dec.prenet1.W").n, Wa=self.lstm_w["dec.att_lstm.W"], Wq=wq.n,
                W1=self.lstm_w["dec.lstm1.W"], W2=self.lstm_w["dec.lstm2.W"], Wkvq=eng.W(pre + ".kvq.W").n, Wot=self.Wot_k[0],
                Wout=self.out_w, bp0=P["dec.prenet0.b"], bp1=P["dec.prenet1.b"], ba=P["dec.att_lstm.b"], b1l=P["dec.lstm1.b"],
                b2l=P["dec.lstm2.b"], bkvq=P[pre + ".kvq.b"], bot=self.bot[0], bout=P["dec.out.b"],
                locF=P["dec.att1.F"], locFb=P["dec.att1.bF"], locU=P["dec.att1.U"], v1=P["dec.att1.v"], b1=P["dec.att1.b"],
                v2=P["dec.att2.v"], lengths=self.lengths, keys1=self.keys1, values1=self.values1, keys2=self.keys2,
                values2=self.values2, ca=ca, ha=ha, c1=c1, h1=h1, c2=c2, h2=h2, a_state=self.a_state,
                alpha_state=self.alpha_state, ctx=self.ctx, yout=self.yout, tin=self.tin, align1=self.al1, align2=self.al2,
                kvq=self.kvqs[0], part=self._mega_part, ctab=self.ctab, step=self.steps2, flag=None if teacher else self.flag,
                err=self._mega_err, **({} if self._fb is None else dict(Wfh=self._fb["hi"], Wfl=self._fb["lo"], bfb=self._fb["bias"])))
            assert ops.dec_mega_supported(self.mega)
            self.kernel_launches = 1          # per K steps
        self.refresh_folded()
        if self.mega is not None:
            # One throw-away launch (zero memories, discarded by the reset() of the first utterance), as the graph path below runs its
            # step once outside the capture: module load and the cold start of the 32 persistent workgroups stay out of the first
            # utterance.  r5: one run in ~30 fresh processes had the FIRST utterance of the first session off by 7e-3 in mel (bar
            # 2.2e-3; every other run of it is bit-identical, 24 dedicated fresh-box trials did not reproduce it) - a start-up race we
            # have not located; later utterances never showed it.
            if __import__("os").environ.get("SATT_DECODE_NO_WARMUP") != "1":      # (the switch: tools/decode_stress.py measures the cold case)
                self.reset()
                self.lengths.fill_(Ti)
                self.replay(self.K)
                torch.cuda.synchronize()
                self.check()
            return
        if use_graph:
            self.reset()
            self.lengths.fill_(Ti)
            self.run_step()                                  # first launches outside the capture (module load, attributes)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(self.K):
                    self.run_step()
            self.graph = g

    @staticmethod
    def _fuse_chains(L):
        """one or two short plain Dense launches whose only consumer is the FIRST input segment of the next launch (pre-net 0 ->
        pre-net 1 -> attention LSTM; folded output transform -> mel | stop projection) ride in that launch as its prologue
        (csrc/decode.hip dec_chain_k): every workgroup recomputes them - cheaper than the ~5 us of a dependent launch each"""
        def plain(a):
            return a.lstm_H == 0 and a.nseg == 1 and a.y_ss == 0 and bool(a.Wb) and a.N <= 256 and a.k[0] <= 256 and a.ldw % 4 == 0

        def feeds(a, b):
            return b.x[0] == a.y and b.x_bs[0] == a.y_bs and b.x_ss[0] == 0 and b.x_ps[0] == 0 and b.k[0] == a.N

        # (the kernel takes two chained layers; measured, the second one costs as much as the launch it saves: B=1 57.9 vs 57.3 us
        # per step, B=8 73.6 vs 70.4 - one layer is chained, the pre-net's first layer keeps its own launch)
        max_pre = DecodeSession.MAX_CHAIN
        out, i = [], 0
        while i < len(L):
            run = []
            j = i
            while j < len(L) and L[j][0] is ops.dec_linear and plain(L[j][1]) and (not run or feeds(run[-1], L[j][1])):
                run.append(L[j][1]); j += 1
            # the consumer: the launch behind the run if it takes the run's output, else the run's own last layer
            if run and j < len(L) and L[j][0] is ops.dec_linear and feeds(run[-1], L[j][1]) and bool(L[j][1].Wb) and L[j][1].ldw % 4 == 0:
                main, pre, nxt = L[j][1], run, j + 1
            elif len(run) >= 2:
                main, pre, nxt = run[-1], run[:-1], j
            else:
                out.append(L[i]); i += 1
                continue
            lead, pre = pre[:-max_pre], pre[-max_pre:]
            out.extend((ops.dec_linear, a) for a in lead)
            fused = type(main).from_buffer_copy(main)
            src = [q for q in pre if q.step_out and q.step_out == main.step]
            if src:
                # a launch must not read a counter word it writes: the pre-layer that copies the counter (step_add == 0) makes
                # its own source word the same value - the main layer reads that one
                assert src[0].step_add == 0
                fused.step = src[0].step
            parts = list(pre) + [main]
            out.append((lambda t: ops.dec_linear_chain(t[0], t[1]) or [ops.dec_linear(a) for a in t[2]], (list(pre), fused, parts)))
            i = nxt
        return out

    @staticmethod
    def _fuse_pairs(L):
        """a plain Dense launch whose only consumer is the next plain Dense launch becomes one launch of the two-layer kernel
        (csrc/decode.hip dec_linear2_k: a dependent launch costs ~5 us whatever it does) where that kernel takes the pair"""
        out, i = [], 0
        while i < len(L):
            fn, a = L[i]
            if fn is ops.dec_linear and i + 1 < len(L) and L[i + 1][0] is ops.dec_linear:
                b = L[i + 1][1]
                chained = (a.lstm_H == 0 and b.lstm_H == 0 and a.nseg == 1 and b.nseg == 1 and b.x[0] == a.y and
                           b.x_bs[0] == a.y_bs and a.y_ss == 0 and b.x_ss[0] == 0 and b.x_ps[0] == 0 and b.k[0] == a.N and
                           bool(a.Wb) and bool(b.Wb) and a.N <= 256 and b.N <= 256 and a.k[0] <= 256 and a.ldw % 4 == 0 and b.ldw % 4 == 0
                           # one workgroup per sample: below 4 samples the single CU's load bandwidth costs what the saved launch gains
                           # (B = 1: 63.7 vs 62.8 us per step; B = 8: 91 vs 102)
                           and a.B >= 4)
                if chained:
                    out.append((lambda pair: ops.dec_linear2(pair[0], pair[1]) or (ops.dec_linear(pair[0]), ops.dec_linear(pair[1])),
                                (a, b)))
                    i += 2
                    continue
            out.append((fn, a))
            i += 1
        return out

    # the persistent kernel where it applies (csrc/decode_mega2.hip); False / SATT_DECODE_MEGA=0: hipGraph of launch-per-layer steps
    MEGA = __import__("os").environ.get("SATT_DECODE_MEGA", "1") != "0"
    MEGA_MAX_B = 2      # the kernel takes B <= 2 (20.6 / 30.4 us per step at B = 1 / 2; the graph path: 57 / 59 us); tests lower it
    # decoder steps per launch of the persistent kernel (at least; see __init__).  The kernel leaves its step loop at the stop token by
    # itself, so a long launch costs nothing past the token; the launch prologue (weights into registers, tables into LDS, placement
    # handshake: ~10 us) is what the length amortises - r6, B = 1, 200 steps: 32 -> 14.2 us per step, 64 -> 13.8, 128 -> 13.5, 224 -> 13.4
    MEGA_STEPS = 128
    MEGA_FOLD_FEEDBACK = __import__("os").environ.get("SATT_DECODE_FOLD_FEEDBACK", "1") != "0"     # projection -> pre-net 0 folded (csrc/decode_mega2.hip)
    FUSE = True         # chain short Dense launches into their consumers (csrc/decode.hip dec_chain_k); tests switch it off
    MAX_CHAIN = 1       # layers chained in front of a consumer (the kernel takes up to 2)

    def refresh_folded(self):
        """weights derived from the parameters (cheap, redone per utterance: the parameters may have been updated)"""
        P = self.eng.P
        for n, dst in self.lstm_w.items():      # column gate * H + u  ->  (u / 8) * 32 + gate * 8 + u % 8 (a copy, no arithmetic)
            K4, H = P[n].shape[0], P[n].shape[1] // 4
            dst.view(K4, H // 8, 4, 8).copy_(P[n].view(K4, 4, H // 8, 8).permute(0, 2, 1, 3))
        self.out_w.copy_(P["dec.out.W"])
        if self.kvq is not None:
            prec = ops.get_precision()
            ops.set_precision("f32")
            from .params import sa_prefix
            try:
                for h, (Wot, bot) in enumerate(zip(self.Wot, self.bot)):
                    pre = sa_prefix("dec.sa", h)
                    ops.linear(P[pre + ".o.W"], P[pre + ".t.W"], None, Wot)
                    ops.linear(P[pre + ".o.b"].view(1, -1), P[pre + ".t.W"], P[pre + ".t.b"], bot)
            finally:
                ops.set_precision(prec)
            for Wk, Wot in zip(self.Wot_k, self.Wot):
                if Wk is not Wot:
                    Wk.copy_(Wot)
        fb = getattr(self, "_fb", None)
        if fb is not None:      # Wf = Wout[:, fed columns] Wp0 from the bf16 weights the unfolded step multiplies with, exact-fp32 product
            c = self.eng.cfg
            feed, NO = c.num_mels * c.n_feed_frame, c.num_mels * c.r + 1
            c0 = NO - 1 - feed
            fb["a"].copy_(self.out_w[:, c0:c0 + feed])                       # (casts: bf16 -> fp32)
            fb["b"].copy_(self.eng.W("dec.prenet0.W").n)
            prec = ops.get_precision()
            ops.set_precision("f32")
            try:
                ops.linear(fb["a"], fb["b"], None, fb["w"])
                ops.linear(P["dec.out.b"][c0:c0 + feed].view(1, -1), fb["b"], P["dec.prenet0.b"], fb["bias"])
            finally:
                ops.set_precision(prec)
            ops.to_bf16(fb["w"], fb["hi"])
            fb["t"].copy_(fb["hi"])
            ops.axpby(fb["w"], fb["t"], 1.0, -1.0)                           # t = Wf - hi
            ops.to_bf16(fb["t"], fb["lo"])

    def build_context_tables(self):
        """per utterance, after the memories are in place: values W_c in fp32 (the bf16-rounded, regrouped weights of the step)"""
        if self.mega is None or self.ctab is None:
            return
        c = self.eng.cfg
        A, D, V1, V2, P1 = c.att_rnn_units, c.dec_units, c.cbhg_out_units, c.sa_units, c.dec_prenet[1]
        W1, Wa = self.lstm_w["dec.lstm1.W"], self.lstm_w["dec.att_lstm.W"]
        rows = ((W1, A, V1), (W1, A + V1, V2), (Wa, P1, V1), (Wa, P1 + V1, V2))
        for q, (W, r0, n) in enumerate(rows):
            self._ctw[q].copy_(W[r0:r0 + n])
            x = self.values1 if q % 2 == 0 else self.values2
            ops.gemm(x.shape[0], 4 * D, n, x, n, self._ctw[q], 4 * D, 1, self.ctab[:, q * 4 * D:], 16 * D, prec=ops.PREC_F32)

    def run_step(self):
        for fn, prm in self.launches:
            fn(prm)

    def replay(self, nsteps=None):
        """K decoder steps (the persistent kernel: `nsteps` <= K of them - a ragged last launch): one launch of the persistent
        kernel, or one replay of the captured graph"""
        if self.mega is not None:
            ops.dec_mega(self.mega, self.K if nsteps is None else max(1, min(self.K, int(nsteps))))
        else:
            self.graph.replay()

    def check(self):
        """host-synchronous: raise if an exchange of the persistent kernel timed out (sticky word)"""
        if self.mega is not None and int(self._mega_err.item()):
            raise SattError("decode: an exchange of the persistent step kernel timed out")

    def reset(self):
        """recurrent state of a new utterance (zeros; alpha_0 = onehot(0): modules/forward_attention.py:128-136)"""
        self.steps2.zero_(); self.flag.zero_()
        if self.mega is not None:
            self._mega_part.zero_()        # exchange granules carry step + 1 as their tag: a new utterance starts from untagged ones
        for t in self.states:
            t.zero_()
        self.ctx.zero_(); self.a_state.zero_(); self.alpha_state.zero_()
        self.alpha_state[0, :, 0] = 1.0
        self.yout[:, 0].zero_()


def infer(eng, source, source_length, max_steps=None, teacher=None, speaker_id=None, min_steps=10, stop_threshold=0.5,
          check_every=8, teacher_alignments=None, use_graph=True, dropout_seed=None, encoder_outputs=None, speaker_embed=None):
    """eng: Engine.  source int64 [B,Ti], source_length int64 [B] (device tensors or array-likes).
    teacher=None: free running, at most max_steps decoder steps, stops when sigmoid(stop) > stop_threshold for every
    sample and t > min_steps (evaluated on the device every step; the host reads the flag once per graph replay =
    `check_every` steps, one replay behind).
    teacher=[B,Tm,num_mels]: inputs from the ground truth (validation pass), exactly Tm/r steps.
    teacher_alignments=(a1, a2), each [B,T,Ti] with T >= the number of steps (a2 = None for the single-source model):
    forced-alignment mode (use_forced_alignment_mode: modules/teacher_forcing_attention.py:13-78, models/models.py:411-428)
    - the mechanisms return the given alignment of the step; contexts and alignment histories follow them.
    use_graph=False issues the same kernels step by step without capturing them (debugging).
    dropout_seed (apply_dropout_on_inference only): seed of this call's pre-net dropout masks.  None: the engine's seed word
    plus a per-call increment, so repeated synthesis of one utterance draws different masks (as the reference's stateful TF
    RNG does); an int pins the masks (parity tests; int(eng.seed) = the masks of Engine.forward(training=False)).
    Returns dict(mel [B,T*r,num_mels], stop [B,T,1], alignment1 [B,T,Ti], alignment2 [B,T,Ti], steps=T,
    lstm_out, sa_out)."""
    c, P, dev = eng.cfg, eng.P, eng.dev
    f32 = dict(dtype=torch.float32, device=dev)
    batch = {"source": torch.as_tensor(source).to(dev).contiguous(),
             "source_length": torch.as_tensor(source_length).to(dev).contiguous()}
    if c.num_speakers > 0 and speaker_embed is None:
        batch["speaker_id"] = torch.as_tensor(speaker_id).to(dev).contiguous()
    B, Ti = batch["source"].shape
    slen = batch["source_length"]
    nm, r = c.num_mels, c.r
    feed = nm * c.n_feed_frame
    NO = nm * r + 1
    if teacher is not None:
        teacher = torch.as_tensor(teacher, **f32).contiguous()
        Td = teacher.shape[1] // r
    else:
        if not max_steps or max_steps < 1:
            raise SattError("infer: max_steps must be given for free-running decode")
        Td = int(max_steps)
    forced = teacher_alignments is not None
    if forced:
        ta1 = torch.as_tensor(teacher_alignments[0], **f32)
        ta2 = torch.as_tensor(teacher_alignments[1], **f32) if (c.dual and teacher_alignments[1] is not None) else None
        if ta1.dim() != 3 or ta1.shape[0] != B or ta1.shape[2] != Ti or ta1.shape[1] < Td or \
                (c.dual and (ta2 is None or ta2.shape != ta1.shape)):
            raise SattError("infer: teacher_alignments must be [B, T >= steps, Ti] tensors, one per attention source")
    ctx = {"training": False, "batch": batch}
    if encoder_outputs is not None:      # the decoder half on given memories (decoder call contract, modules/module.py)
        lstm_out = torch.as_tensor(encoder_outputs[0], **f32).reshape(B * Ti, c.cbhg_out_units).contiguous()
        sa_out = torch.as_tensor(encoder_outputs[1], **f32).reshape(B * Ti, c.sa_units).contiguous() if c.dual else None
        ctx["enc_align"] = None
    else:
        lstm_out, sa_out = eng._encode(batch, False, ctx)
    key = (B, Ti, Td, teacher is not None, forced, int(min_steps), float(stop_threshold), int(check_every), bool(use_graph),
           ops.get_precision(), DecodeSession.FUSE, DecodeSession.MAX_CHAIN, DecodeSession.MEGA, DecodeSession.MEGA_MAX_B,
           DecodeSession.MEGA_STEPS, DecodeSession.MEGA_FOLD_FEEDBACK)
    cache = eng.__dict__.setdefault("_decode_sessions", {})
    ses = cache.get(key)
    if ses is None:         # (the kernels read the parameters in place: an optimiser step does not invalidate a session)
        if len(cache) >= 8:
            cache.clear()
        ses = cache[key] = DecodeSession(eng, B, Ti, Td, teacher is not None, forced, min_steps, stop_threshold, check_every,
                                         use_graph)
    # ---- memories (same as Engine.forward): values = memory * seq_mask, keys = values W_m
    V1, V2 = c.cbhg_out_units, c.sa_units
    ses.lengths.copy_(slen)
    ops.seq_mask(lstm_out, slen, ses.values1, B, Ti, V1)
    ops.linear(ses.values1, P["dec.att1.Wm"], None, ses.keys1)
    if c.dual:
        ops.seq_mask(sa_out, slen, ses.values2, B, Ti, V2)
        ops.linear(ses.values2, P["dec.att2.Wm"], None, ses.keys2)
    if c.num_speakers > 0:      # multi-speaker pre-net term (constant over time): softsign(emb[speaker] Ws + bs)
        semb = torch.empty(B, c.speaker_dim, **f32)
        if speaker_embed is not None:
            semb.copy_(torch.as_tensor(speaker_embed, **f32).reshape(B, c.speaker_dim))
        else:
            ops.embedding_fwd(batch["speaker_id"], P["speaker_embedding"], semb, offset=c.speaker_offset)
        ops.linear(semb, P["dec.prenet0.Ws"], P["dec.prenet0.bs"], ses.sproj, act=ACT_SOFTSIGN)
    if teacher is not None:
        tg = teacher.view(B, Td, nm * r)
        ses.tin[:, 0].zero_()
        ses.tin[:, 1:Td] = tg[:, :Td - 1, nm * r - feed:]
    if forced:
        ses.teach1[:, :Td] = ta1[:, :Td]
        if ses.teach2 is not None:
            ses.teach2[:, :Td] = ta2[:, :Td]
    ses.refresh_folded()
    ses.build_context_tables()
    ses.reset()
    if c.apply_dropout_on_inference:
        if dropout_seed is None:
            n = eng.__dict__["_infer_calls"] = eng.__dict__.get("_infer_calls", 0) + 1
            ses.drop_seed.copy_(eng.seed)
            ses.drop_seed += (n * 0x3C6EF35F) % (1 << 31)           # device add (int32 wrap-around is fine for a hash seed)
        else:
            ses.drop_seed.fill_(int(dropout_seed))
    K = ses.K
    steps = Td
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_a.record()
    if ses.graph is None and ses.mega is None:
        for t in range(Td):
            ses.run_step()
            if teacher is None and t > min_steps and (t % K == 0 or t == Td - 1):
                f = int(ses.flag.item())
                if f:
                    steps = f
                    break
    else:
        pend = []               # (event after replay i, pinned copy of the flag): read one replay behind
        nrep = ses.Tdp // K
        hostbuf = torch.empty(nrep, dtype=torch.int32, pin_memory=True) if teacher is None else None
        for i in range(nrep):
            ses.replay(Td - i * K)
            if teacher is not None or (i + 1) * K <= min_steps:
                continue
            host = hostbuf[i:i + 1]
            host.copy_(ses.flag, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
            pend.append((ev, host))
            if len(pend) > 1:
                ev0, h0 = pend.pop(0)
                ev0.synchronize()
                if int(h0[0]):
                    break
        torch.cuda.current_stream().synchronize()
        f = int(ses.flag.item()) if teacher is None else 0
        if f:
            steps = min(f, Td)
        ses.check()
    ev_b.record(); ev_b.synchronize()
    y = ses.yout[:, 1:steps + 1]
    yout = y.reshape(B * steps, NO) if steps == Td else None
    return dict(yout=yout, mel=y[:, :, :NO - 1].reshape(B, steps * r, nm), stop=y[:, :, NO - 1:].contiguous(),
                alignment1=ses.al1[:, :steps].clone(), alignment2=ses.al2[:, :steps].clone() if c.dual else None,
                steps=steps, decode_ms=ev_a.elapsed_time(ev_b), lstm_out=lstm_out.view(B, Ti, -1), sa_out=sa_out.view(B, Ti, -1) if c.dual else None,
                enc_alignment=ctx["enc_align"].view(B, c.sa_heads, Ti, Ti) if (c.dual and ctx.get("enc_align") is not None) else None)


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
