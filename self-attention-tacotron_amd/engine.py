"""Teacher-forced training engine: forward, hand-written backward and fused optimiser of
DualSourceSelfAttentionTacotronModel (reference models/models.py:278-515) and - with sa_units = att2_units =
dec_sa_units = 0 - of the baseline ExtendedTacotronV1Model (models/models.py:20-226: ZoneoutEncoderV1 + ExtendedDecoder
v2, one attention source, no self-attention blocks), every arithmetic op a HIP kernel
behind the C-ABI (ops.py -> libsatt_hip.so).  torch supplies device memory, streams and torch.distributed only.

Layer-wise wavefront: under teacher forcing the attention RNN never depends on LSTM1/LSTM2, so the
loop is split into three persistent recurrent kernels (attention RNN, LSTM1, LSTM2) whose input projections are
hoisted into large batched MFMA GEMMs over all B*Td steps.
"""
import math
import os

import torch

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_SOFTSIGN, ACT_TANH, Drop
from .params import ModelConfig, init_params, layout, sa_prefix, sa_prefixes

# dropout / zoneout stream ids (== oracle/rng.py; the mask function lives in csrc/common.h)
S_ENC_PRENET0, S_ENC_PRENET1 = 1, 2
S_ENC_FW_C, S_ENC_FW_H, S_ENC_BW_C, S_ENC_BW_H = 3, 4, 5, 6
S_ENC_SA = 7
HOP_STREAM = 64     # dropout stream of hop h of a SelfAttentionTransformer stack = the stack's stream + HOP_STREAM * h
S_DEC_PRENET0, S_DEC_PRENET1 = 8, 9
S_ATT_C, S_ATT_H, S_L1_C, S_L1_H, S_L2_C, S_L2_H = 10, 11, 12, 13, 14, 15
S_DEC_SA = 16
S_POSTNET0 = 17   # + layer index


class Engine:
    def __init__(self, cfg: ModelConfig, device="cuda", param_seed=0, params=None, rng_seed=0,
                 lr0=5e-4, decay=True, step_factor=1.0, b1=0.9, b2=0.999, eps=1e-8, clip=1.0, loss_type="l1"):
        self.cfg = cfg
        if cfg.attention not in ("forward", "location_sensitive"):
            from .modules.attentions import UnsupportedConfiguration
            raise UnsupportedConfiguration("attention=%s: the attention-RNN kernels implement forward and "
                                           "location_sensitive" % cfg.attention)
        self.dev = torch.device(device)
        if self.dev.type == "cuda":
            ops.bind_device(self.dev)
        self.layout, self.nparam = layout(cfg)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.flat = torch.zeros(self.nparam, **f32)
        self.grad = torch.zeros(self.nparam, **f32)
        self.m = torch.zeros(self.nparam, **f32)
        self.v = torch.zeros(self.nparam, **f32)
        self.P = {k: self.flat[o:o + math.prod(s)].view(s) for k, (o, s) in self.layout.items()}
        self.G = {k: self.grad[o:o + math.prod(s)].view(s) for k, (o, s) in self.layout.items()}
        init = params if params is not None else init_params(cfg, param_seed)
        for k, a in init.items():
            self.P[k].copy_(torch.as_tensor(a, dtype=torch.float32))
        self.enc_end = self.layout["dec.prenet0.W"][0] if cfg.num_speakers == 0 else self.layout["speaker_embedding"][0]
        self.enc_mid = self.layout["enc.proj1.W"][0]        # DP bucket boundary inside the encoder (see train_step)
        # BatchNorm moving statistics (buffers, not parameters)
        nb = cfg.max_filter_width * cfg.conv_channels
        self.bn = {n: (torch.zeros(c, **f32), torch.ones(c, **f32))
                   for n, c in (("bank", nb), ("proj1", cfg.proj1), ("proj2", cfg.proj2))}
        if cfg.use_postnet_v2:
            for n in range(cfg.num_postnet_v2_layers):
                self.bn[f"postnet{n}"] = (torch.zeros(cfg.postnet_v2_out_channels, **f32),
                                          torch.ones(cfg.postnet_v2_out_channels, **f32))
        self.post_losses = torch.zeros(3, **f32)
        # L2 regularisation of the baseline model (reference modules/regularizers.py, models/models.py:109-114)
        self.reg_loss = torch.zeros(1, **f32)
        self._l2_table, self._l2_n = None, 0
        if cfg.l2_weight > 0:
            from .params import l2_regularized
            segs = [(self.layout[n][0], math.prod(self.layout[n][1])) for n in l2_regularized(cfg)]
            self._l2_table = torch.tensor(segs, dtype=torch.int64, device=self.dev).contiguous()
            self._l2_n = len(segs)
        self._loss_ws2 = torch.zeros(4, **f32)
        self.opt_state = ops.opt_state(self.dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.seed = torch.full((1,), rng_seed, dtype=torch.int32, device=self.dev)
        self.hyper = dict(lr0=lr0, decay=decay, step_factor=step_factor, b1=b1, b2=b2, eps=eps, clip=clip)
        self.loss_l2 = (loss_type == "mse")
        self.losses = torch.zeros(3, **f32)
        self._loss_ws = torch.zeros(8, **f32)
        # bf16 shadows of every matrix / conv parameter for the large-tile GEMMs (csrc/gemm_tile.hip): same offsets as the
        # flat fp32 buffer; st = per-tap transpose (forward products), sn = plain cast (input gradients)
        self.st_flat = torch.zeros(self.nparam, dtype=torch.bfloat16, device=self.dev)
        self.sn_flat = torch.zeros(self.nparam, dtype=torch.bfloat16, device=self.dev)
        tab, self._wref = [], {}
        for k, (o, shp) in self.layout.items():
            if len(shp) < 2:
                continue
            taps, R, Cc = (1, shp[0], shp[1]) if len(shp) == 2 else shp
            tab += [o, taps, R, Cc]
            n = math.prod(shp)
            tshape = (Cc, R) if len(shp) == 2 else (taps, Cc, R)
            self._wref[k] = ops.Weight(self.P[k], self.st_flat[o:o + n].view(tshape), self.sn_flat[o:o + n].view(shp))
        self._shadow_table = torch.tensor(tab, dtype=torch.int64, device=self.dev)
        self._shadow_count = len(tab) // 4
        self.shadow = {}
        self.refresh_shadows()

    # ------------------------------------------------------------------ bf16 shadows of the recurrent weights
    def _shadow(self, name, src):
        rows, cols = src.shape
        for tr in (False, True):
            key = name + (".T" if tr else "")
            if key not in self.shadow:
                self.shadow[key] = torch.empty((cols, rows) if tr else (rows, cols), dtype=torch.bfloat16,
                                               device=self.dev)
            ops.to_bf16(src, self.shadow[key], transpose=tr)

    def W(self, name):
        """GEMM operand of parameter `name`: fp32 master view + bf16 shadows (ops.Weight)"""
        return self._wref[name]

    def refresh_shadows(self, after_gemm_shadows=None):
        """bf16 copies of the parameters, re-made after every update.  after_gemm_shadows(): called once the shadows the GEMMs
        read exist (one launch); the rest - the recurrent kernels' operand layouts - is first needed by the encoder LSTM."""
        c, P = self.cfg, self.P
        ops.shadow_pack(self.flat, self._shadow_table, self._shadow_count, self.st_flat, self.sn_flat)
        if after_gemm_shadows is not None:
            after_gemm_shadows()
        used = [k for k in getattr(self, "_pack_cache", {})]       # cluster sizes of the last step: re-packed right away
        self._pack_cache = {}       # per-cluster-size weight slices are re-packed after every update
        H = c.cbhg_out_units // 2
        if "enc.Wh" not in self.shadow:
            self.shadow["enc.Wh"] = torch.empty(2, H, 4 * H, dtype=torch.bfloat16, device=self.dev)
            self.shadow["enc.Wh.T"] = torch.empty(2, 4 * H, H, dtype=torch.bfloat16, device=self.dev)
        for d, n in enumerate(("fw", "bw")):
            ops.to_bf16(P[f"enc.lstm_{n}.W"][H:], self.shadow["enc.Wh"][d], False)
            ops.to_bf16(P[f"enc.lstm_{n}.W"][H:], self.shadow["enc.Wh.T"][d], True)
        pn = c.dec_prenet[-1]
        self._shadow("att.Wrec", P["dec.att_lstm.W"][pn:])
        self._shadow("att.Wq", P["dec.att.Wq"])
        A = c.att_rnn_units
        self._shadow("l1.Wh", P["dec.lstm1.W"][A + c.ctx_dim:])
        self._shadow("l2.Wh", P["dec.lstm2.W"][c.dec_units:])
        if self._out_pad():             # output projection with zero-padded rows (input-gradient GEMM, see backward())
            NO = P["dec.out.W"].shape[1]
            if "out.Wn" not in self.shadow:
                self.shadow["out.Wn"] = torch.zeros(c.out_in, NO + self._out_pad(), dtype=torch.bfloat16, device=self.dev)
            self.shadow["out.Wn"][:, :NO].copy_(P["dec.out.W"])
        # register-order packs of the cluster kernels for the cluster sizes the last step used: here (on the stream this
        # runs on, after the update) instead of between the encoder and the decoder loop of the next step
        for k in used:
            if isinstance(k, tuple) and k[0] == "lstm":
                self.lstm_cluster_packs(k[1])
            elif isinstance(k, tuple) and k[0] == "lstm_in":
                self.lstm_cluster_in_packs(k[1])
            elif isinstance(k, int):
                self._pack_cache[k] = ops.attn_cluster_pack(P["dec.att_lstm.W"][c.dec_prenet[-1]:], A, k)
            elif isinstance(k, tuple) and k[0] == "fold":
                self._pack_cache[k] = ops.attn_cluster_pack(P["dec.att_lstm.W"][c.dec_prenet[-1] + c.cbhg_out_units:], A, k[1])
        self._refresh_folded()

    def _refresh_folded(self):
        """The transformer tail applies two Dense layers back to back with nothing in between (output projection of the
        multi-head attention, then SelfAttentionTransformer's Dense under the tanh: modules/self_attention.py:119-128,
        modules/module.py:363-371): tanh((o Wo + bo) Wt + bt) = tanh(o Wot + bot) with Wot = Wo Wt, bot = bo Wt + bt.  The
        folded pair (fp32, exact-fp32 product, + bf16 shadows) is re-made after every update; the forward pass and the
        input-gradient chain then cost ONE GEMM instead of two - the two weight gradients keep their own (side-stream) work."""
        c, P = self.cfg, self.P
        if not hasattr(self, "_folded"):
            self._folded = {}
        prec = ops.get_precision()
        ops.set_precision("f32")
        try:
            for prefix, D, _ in sa_prefixes(c):          # every hop of both stacks
                if prefix not in self._folded:
                    f32 = dict(dtype=torch.float32, device=self.dev)
                    b16 = dict(dtype=torch.bfloat16, device=self.dev)
                    self._folded[prefix] = (ops.Weight(torch.empty(D, D, **f32), torch.empty(D, D, **b16), torch.empty(D, D, **b16)),
                                            torch.empty(1, D, **f32))
                Wd, bd = self._folded[prefix]
                ops.linear(P[prefix + ".o.W"], P[prefix + ".t.W"], None, Wd.w)
                ops.linear(P[prefix + ".o.b"].view(1, -1), P[prefix + ".t.W"], P[prefix + ".t.b"], bd)
                ops.to_bf16(Wd.w, Wd.t, transpose=True)
                ops.to_bf16(Wd.w, Wd.n, transpose=False)
        finally:
            ops.set_precision(prec)

    # ------------------------------------------------------------------ helpers
    _keep = None   # during backward: every temporary stays alive until the side streams have been joined

    def dec_prenet_rate(self, training, plain):
        """dropout rate of a decoder pre-net layer: plain PreNet layers stay stochastic outside training when
        apply_dropout_on_inference is set (modules/module.py:564-577); MultiSpeakerPreNet never does (:569-570)"""
        on = training or (plain and self.cfg.apply_dropout_on_inference)
        return self.cfg.dec_prenet_drop if on else 0.0

    def _bn_fused_state(self, name, rows, Cc):
        """workspace + barrier words of the one-launch BatchNorm of layer `name` (persistent: the words are zero between launches)"""
        cache = self.__dict__.setdefault("_bn_states", {})
        key = (name, rows, Cc)
        if key not in cache:
            cache[key] = ops.bn_fused_state(rows, Cc, self.dev)
        return cache[key]

    def _cu_count(self):
        if self.__dict__.get("_ncu") is None:
            self._ncu = torch.cuda.get_device_properties(self.dev).multi_processor_count if torch.cuda.is_available() else 256
        return self._ncu

    def _layers_fit_side_by_side(self, ap, Ca, vw1, B, Td, D, Cn):
        """May the attention cluster kernel and the LSTM cluster launches of the layer pipeline be in flight TOGETHER?  Every member
        of a cluster spins (bounded) for its peers, so a launch whose workgroups cannot all become resident - because workgroups of
        ANOTHER spinning launch hold the CUs they need - ends in hand-off time-outs.  Footprints (satt_*_cluster_residency:
        hipOccupancyMaxActiveBlocksPerMultiprocessor for the kernel and LDS size of the launch, CU count of this device):
          * attention (forward and backward kernel, the larger): ceil(workgroups / workgroups per CU) CUs of its own - it takes the
            whole register file of a CU, nothing shares a CU with it;
          * each LSTM cluster launch that can be in flight at the same time (one with lstm_one_stream, else two): the dispatcher
            SPREADS workgroups over the CUs, so every workgroup is charged a CU of its own even where two would fit.
            r6 tried the calculator's two per CU (`lstm_cu_charge = "packed"`, forward kernel pinned to 128 registers): the B = 33
            cliff goes (B = 33 / 36 / 40 at Tm = 400: 6.03 / 5.66 / 5.40 -> 4.69 / 4.51 / 4.55 ms per step) and the FORWARD pipeline
            never failed up to B = 42 (zero CUs to spare), but the BACKWARD pipeline times out intermittently - always at B = 42,
            1 in ~3 runs at B = 40, 1 in 12 at B = 38, once at B = 36 - also with the LSTM stream held back until every attention
            workgroup is resident (`attention_first`).  There the LSTM launches share their CUs with the deferred attention
            gradients and the weight-gradient GEMMs; which of them keeps a packed LSTM launch from becoming resident as a whole
            is not located (profiles/r06_residency_packed.txt).  An intermittent time-out is a skipped update: not shipped.
        Side by side only if the sum stays within the device.  Without a device to ask (CPU import): the r4 rule."""
        key = (B, ap.Ti, Ca, Cn, D, vw1 is not None, bool(ap.saf), self.lstm_one_stream, self.lstm_cu_charge)
        cache = self.__dict__.setdefault("_fit_cache", {})
        if key not in cache:
            rf = ops.attn_cluster_residency(ap, Ca, False, vw1=vw1)
            rb = ops.attn_cluster_residency(ap, Ca, True)
            lf = ops.lstm_cluster_residency(B, Td, D, Cn, False)
            lb = ops.lstm_cluster_residency(B, Td, D, Cn, True)
            if None in (rf, rb, lf, lb):
                cache[key] = dict(fits=B * (Ca + Cn) <= self._cu_count(), attention_first=False)
            else:
                cus = rf[2]
                attn = max(-(-r[0] // max(r[1], 1)) for r in (rf, rb))
                spread = (1 if self.lstm_one_stream else 2) * min(max(lf[0], lb[0]), cus)
                lstm = max(-(-r[0] // max(r[1], 1)) for r in (lf, lb)) if (self.lstm_one_stream and self.lstm_cu_charge == "packed") else spread
                fits = all(r[0] <= r[1] * r[2] for r in (rf, rb, lf, lb)) and attn + lstm <= cus
                # beyond the r5 rule (one CU per LSTM workgroup) the launch ORDER matters: the attention kernel must be resident as
                # a whole before an LSTM launch spreads over the CUs it needs (backward(): the LSTM stream waits for it)
                cache[key] = dict(attention=(rf, rb), lstm=(lf, lb), attention_cus=attn, lstm_cus=lstm, cus=cus, fits=fits,
                                  attention_first=bool(fits and attn + spread > cus))
        self.residency = cache[key]
        return cache[key]["fits"]

    def _out_pad(self):
        """pad columns behind the [mel | stop] rows of the output projection"""
        return (-(self.cfg.num_mels * self.cfg.r + 1)) % 8

    def _e(self, *shape, dtype=torch.float32):
        t = torch.empty(*shape, dtype=dtype, device=self.dev)
        if self._keep is not None:
            self._keep.append(t)
        return t

    class _Timed:
        """HIP-event bracket around one kernel launch on the CURRENT stream (bench.py's live roofline timing)."""

        def __init__(self, eng, name):
            self.eng, self.name = eng, name

        def __enter__(self):
            self.on = self.eng.timing is not None and (self.eng.timing_names is None or self.name in self.eng.timing_names)
            if self.on:
                self.a = torch.cuda.Event(enable_timing=True)
                self.b = torch.cuda.Event(enable_timing=True)
                self.a.record()

        def __exit__(self, *exc):
            if self.on:
                self.b.record()
                self.eng.timing.setdefault(self.name, []).append((self.a, self.b))

    timing = None   # set to {} to collect (start, end) event pairs per kernel name
    timing_names = None   # restrict the brackets to these names (every bracket costs two event packets on its stream)
    use_clusters = True   # LDS-resident multi-workgroup recurrent kernels where shapes allow
    pg_lds_pad = 96 * 1024   # dynamic-LDS pad of the deferred attention gradients (keeps them off the attention CUs)
    pipeline_chunks = 8   # time chunks of the attention-RNN -> LSTM1 -> LSTM2 stream pipeline (1 = off)
    pipeline_growth = 1.4   # ratio of consecutive tail chunks (profiles/r04_chunk_sweep.txt: 1.3 - 1.6 within 0.05 ms, larger ratios lose: the
    #                         LSTM chain of a chunk - two layers + three GEMMs, ~100 us for 16 steps - must end before the attention
    #                         kernel has produced the next, smaller chunk or the drain behind the loop grows)
    pipeline_tail = (6, 3)  # (number of geometrically shrinking tail chunks, smallest = Td / (this * chunks))
    # the forward pipeline's own tail (None: the same): the two directions chunk the steps independently
    pipeline_tail_fwd = tuple(int(v) for v in os.environ["SATT_TAIL_FWD"].split(",")) if os.environ.get("SATT_TAIL_FWD") else None
    pipeline_kvq = True              # decoder self-attention K|V|Q projection chunk by chunk on the LSTM2 stream
    # forward chunks of at most this many steps form their LSTM input projections inside the LSTM cluster launch (bf16 mode;
    # csrc/lstm_cluster.hip, fused input projection): the chain behind the attention kernel's last steps loses two GEMM launches
    fuse_xg_steps = int(os.environ.get("SATT_FUSE_XG_STEPS", "32"))
    single_launch_attention = True   # attention kernels span all pipeline chunks and signal chunk ends (see forward())
    save_attention_factors = True    # (with fold_context) energy-derivative factors saved by the forward kernel for the backward one
    fold_context = True              # first-source context folded into the recurrent product where the kernel offers it
    fused_highway = True             # the encoder's highway stack in one launch per direction (bf16 mode, 128 units)
    _keep_fwd = None
    _join = None
    _side = None

    dp_buckets = int(os.environ.get("SATT_DP_BUCKETS", "3"))   # gradient all-reduce buckets per step (2: encoder in one piece)
    overlap_wgrad = True  # weight-gradient GEMMs / bias column sums run on a side stream (never on the dX chain)
    _wg_stream = None

    _wg_rr = None         # streams the weight-gradient work is spread over (round robin); None = [weight-gradient stream]
    _wg_next = 0
    _wg_used = None

    _wg_pending = None    # deferred hand-offs (see _wgrad): launched together behind ONE event record
    wgrad_defer = True    # False: every hand-off records its own event (the round-1 behaviour)

    def _wgrad(self, fn, defer=False):
        """Run fn() (weight-gradient accumulation into self.grad: reads activations / gradients that are never
        overwritten later in the backward pass) on a side stream, ordered after the work issued so far on the current
        stream.  After the recurrent pipeline has drained, its two streams are idle and the work is spread round-robin
        over the three side streams (they map 1:1 onto the remaining hardware queues): the tail of a step is a long
        list of small GEMMs that would otherwise serialise on one queue.
        defer=True: fn is only REGISTERED; _wgrad_flush() launches everything registered so far behind one event.  An event
        record is a marker packet between two dependent kernels of the main stream and costs ~5 us of queue time there
        (tools/event_cost.py) - more than most of the kernels it separates - so the ~26 hand-offs of a backward pass share ~8
        records.  A deferred fn must not depend on Python variables that are rebound later (bind loop variables as defaults)."""
        if not self.overlap_wgrad:
            fn()
            return
        cur = ops.current_stream()
        if defer and self.wgrad_defer:
            if self._wg_pending is None:
                self._wg_pending = []
            self._wg_pending.append((fn, cur))
            return
        self._wgrad_flush()
        self._wgrad_launch([fn], cur)

    def _wgrad_flush(self):
        """launch the deferred hand-offs, each group ordered after the work issued so far on the stream it was registered on
        (one event per registering stream - normally just the main stream)"""
        if self._wg_pending:
            items, self._wg_pending = self._wg_pending, None
            while items:
                cur = items[0][1]
                self._wgrad_launch([f for f, s in items if s == cur], cur)
                items = [(f, s) for f, s in items if s != cur]

    def _wgrad_launch(self, fns, cur):
        if self._wg_stream is None:
            self._wg_stream = self._device_streams(self.dev)[2]
        rr = self._wg_rr or [self._wg_stream]
        ev, waited = None, []
        for fn in fns:
            tgt = rr[self._wg_next % len(rr)]
            if tgt == cur and len(rr) > 1:
                self._wg_next += 1
                tgt = rr[self._wg_next % len(rr)]
            self._wg_next += 1
            if tgt != cur and tgt not in waited:
                if ev is None:
                    ev = torch.cuda.Event(); ev.record(cur)
                tgt.wait_event(ev)
                waited.append(tgt)
            with ops.on_stream(tgt):
                fn()
            if self._wg_used is None:
                self._wg_used = []
            if tgt not in self._wg_used:
                self._wg_used.append(tgt)

    def _wgrad_gather(self, onto):
        """order everything issued through _wgrad so far before the work issued next on stream `onto`"""
        self._wgrad_flush()
        for st in (self._wg_used or []):
            if st != onto:
                ev = torch.cuda.Event(); ev.record(st)
                onto.wait_event(ev)

    def _wgrad_join(self):
        self._wgrad_gather(ops.current_stream())
        self._wg_used = None
        self._wg_rr = None

    def lstm_cluster_in_packs(self, Cn):
        """register-order packs of the INPUT weights of LSTM1 / LSTM2 for the fused input projection of the short forward chunks
        (ops.lstm_cluster_fwd_x), re-packed after every update; None where the input is wider than the kernel takes"""
        key = ("lstm_in", Cn)
        if key not in self._pack_cache:
            c, P = self.cfg, self.P
            n1, D = c.att_rnn_units + c.ctx_dim, c.dec_units
            ok = lambda k: k <= ops.LSTM_CLUSTER_FUSED_KMAX and k % 4 == 0
            self._pack_cache[key] = (ops.lstm_cluster_pack_in(P["dec.lstm1.W"][:n1], D, Cn) if ok(n1) else None,
                                     ops.lstm_cluster_pack_in(P["dec.lstm2.W"][:D], D, Cn) if ok(D) else None)
        return self._pack_cache[key]

    def lstm_cluster_packs(self, Cn):
        """register-order weight packs ((fwd, bwd) of LSTM1, (fwd, bwd) of LSTM2) for cluster size Cn, re-packed from
        the fp32 master weights after every update"""
        key = ("lstm", Cn)
        if key not in self._pack_cache:
            c, P = self.cfg, self.P
            n1 = c.att_rnn_units + c.ctx_dim
            D = c.dec_units
            self._pack_cache[key] = (ops.lstm_cluster_pack(P["dec.lstm1.W"][n1:], D, Cn),
                                     ops.lstm_cluster_pack(P["dec.lstm2.W"][D:], D, Cn))
        return self._pack_cache[key]

    def _chunk_bounds(self, Td, NC, tail=None):
        """time-chunk boundaries of the layer pipeline: equal chunks except that the LAST chunks shrink geometrically
        (the forward pipeline's drain and the backward pipeline's fill) and the FIRST chunk is split once more (the
        backward pipeline's drain: the deferred attention gradients of the last processed chunk run after the loop)."""
        if NC <= 1 or Td < 2 * NC:
            return [(i * Td // NC, (i + 1) * Td // NC) for i in range(NC) if (i + 1) * Td // NC > i * Td // NC]
        sizes = []
        rem = Td
        ntail, tdiv = tail or self.pipeline_tail
        size = fsize = max(min(8, max(1, Td // 8)), Td // (tdiv * NC))    # (a launch per chunk: no chunks of a step or two)
        while len(sizes) < ntail and rem - size > Td // 2:
            sizes.append(size); rem -= size
            fsize *= self.pipeline_growth; size = max(size + 1, int(fsize))
        nb = max(1, NC - len(sizes))
        cuts = [i * rem // nb for i in range(nb + 1)]
        head = max(1, Td // (3 * NC))
        if len(cuts) > 1 and cuts[1] > 2 * head:
            cuts.insert(1, head)
        for sz in reversed(sizes):
            cuts.append(cuts[-1] + sz)
        return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]

    @staticmethod
    def _head_split_row(bounds, chunks_in_suffix, tile):
        """First row of the suffix of the split decoder head (backward()): the tile boundary at or below the start of the
        `chunks_in_suffix`-th last pipeline chunk - every chunk from there on only needs the suffix launch of the fused attention
        backward.  0: no split (the boundary would be row 0)."""
        return bounds[-min(len(bounds), max(1, chunks_in_suffix))][0] // tile * tile

    @staticmethod
    def _backward_pieces(bounds, Td, max_pieces=16, targets=(8, 16, 24, 32, 40)):
        """Pieces [(t0, t1)] of the single-launch backward attention kernel in PROCESSING order (late to early) and,
        per pipeline chunk (same order), the number of pieces up to and including that chunk.  The chunks in the first
        quarter of the sequence - processed last - are cut into pieces, the finer the later they are processed
        (`targets`: piece length of the last chunk, the one before it, ...): the deferred attention gradients of a piece
        start when the piece is done, and what is still missing when the loop ends is on the critical path to the
        encoder backward (r4: the gradient launches lost their fixed cost, so short pieces are cheap)."""
        chunks = list(reversed(bounds))
        cuts = [1] * len(chunks)
        budget = max_pieces - len(chunks)
        for k, ci in enumerate(reversed(range(len(chunks)))):
            b0, b1 = chunks[ci]
            if b0 >= Td // 4 or budget <= 0:
                break
            size = targets[min(k, len(targets) - 1)]
            cuts[ci] = max(1, min((b1 - b0 + size - 1) // size, 1 + budget))
            budget -= cuts[ci] - 1
        pieces, upto = [], []
        for (b0, b1), n in zip(chunks, cuts):
            edges = [b0 + (b1 - b0) * i // n for i in range(n + 1)]
            pieces += [(edges[i], edges[i + 1]) for i in reversed(range(n)) if edges[i + 1] > edges[i]]
            upto.append(len(pieces))
        return pieces, upto

    @staticmethod
    def _merge_leading(chunks, Td):
        """Merge the LEADING entries of [(t0, t1, tag)] (processing order: late to early) while the merged span stays
        within 30 % of the sequence; the merged entry carries the tag of its last member.  Later entries stay as is."""
        merged, merging = [], True
        for (t0, t1, tag) in chunks:
            if merging and merged and (merged[-1][1] - merged[-1][0]) + (t1 - t0) <= max(1, (3 * Td) // 10):
                merged[-1] = (t0, merged[-1][1], tag)          # extend downwards
            else:
                merging = not merged
                merged.append((t0, t1, tag))
        return merged

    def _bank_contiguous(self):
        """the conv-bank weights of widths 1..K lie back to back in the flat buffer (true unless padding intervened)"""
        if getattr(self, "_bank_contig", None) is None:
            c, P = self.cfg, self.P
            base, unit = P["enc.bank1.W"].data_ptr(), 4 * P["enc.bank1.W"].numel()
            self._bank_contig = all(P[f"enc.bank{k}.W"].data_ptr() == base + unit * (k * (k - 1) // 2)
                                    for k in range(1, c.max_filter_width + 1))
        return self._bank_contig

    def _pg(self):
        if getattr(self, "_pg_stream", None) is None:
            self._pg_stream = torch.cuda.Stream(device=self.dev)
        return self._pg_stream

    # The three side streams are PROCESS-WIDE singletons per device.  ROCm multiplexes streams onto 4 hardware queues
    # (the default stream + 3): streams drawn anew for every Engine come from torch's round-robin pool and can land on
    # the hardware queue of the stream that runs the attention kernels - harmless with one launch per chunk (it only
    # serialises), but the single-launch attention kernels WAIT in-kernel for work of the other streams, which must
    # therefore never queue behind them.
    _shared_streams = {}

    # Which hardware queue a stream lands on depends on how many streams the process created before (torch draws them
    # round-robin from a pool; RCCL's process group takes one as well): with a one-rank RCCL group initialised first, the
    # three streams drawn next put the weight-gradient stream on the main stream's queue and the recurrent loops ran 35 %
    # slower (13.4 instead of 10.2 ms per step).  So the streams are SELECTED: pool streams are drawn until three are found
    # that run concurrently with the main stream and with each other (satt_stream_probe, a few ms once per process).
    @classmethod
    def _device_streams(cls, dev):
        key = str(torch.device(dev))
        if key not in cls._shared_streams:
            main = torch.cuda.current_stream(torch.device(dev))
            chosen, spare = [], []
            for _ in range(12):
                st = torch.cuda.Stream(device=dev)
                if all(ops.streams_run_concurrently(o, st, spins=4000) and ops.streams_run_concurrently(st, o, spins=4000)
                       for o in [main] + chosen):
                    chosen.append(st)
                    if len(chosen) == 3:
                        break
                else:
                    spare.append(st)
            # nothing runs concurrently at all (a counter-collecting profiler serialises kernels): any three streams do -
            # forward() / backward() see the same probe result and fall back to one attention launch per chunk
            chosen += spare[:3 - len(chosen)]
            while len(chosen) < 3:
                chosen.append(torch.cuda.Stream(device=dev))
            cls._shared_streams[key] = tuple(chosen)
        return cls._shared_streams[key]

    # Both decoder LSTM layers share ONE stream: next to the attention kernel (one workgroup per CU on half of the chip) there is
    # room for the workgroups of ONE LSTM cluster kernel whatever else is in flight, but two LSTM launches that start dispatching
    # at the same moment (LSTM1 of chunk k, LSTM2 of chunk k+1 - they are released by the same event) could each end up
    # partially resident, every resident member spinning for peers the dispatcher no longer placed: measured at Td = 250 (any
    # Ti, both configurations) as 1.2 s hand-off timeouts at the start of the backward loop, several per step.  The overlap of
    # the two layers that this gives up is bought back by tail chunks that grow by 1.4x instead of 2x (pipeline_growth).
    # (r4: the LSTM cluster kernels now fit two workgroups per CU, i.e. both layers' launches fit the free half of the chip - re-measured
    #  with two streams: within noise of one stream at the default chunking, and with 8-step tail chunks the hand-off time-outs are
    #  back (profiles/r04_chunk_sweep_b.txt).  SATT_LSTM_STREAMS=2 keeps the switch for experiments.)
    lstm_one_stream = os.environ.get("SATT_LSTM_STREAMS", "1") != "2"
    # "spread" (default): one CU per LSTM workgroup; "packed" (r6 experiment, needs the -DSATT_LSTM_FWD_PIN build): the occupancy
    # calculator's two per CU - removes the B = 33 cliff but time-outs intermittently in the backward pipeline (_layers_fit_side_by_side)
    lstm_cu_charge = os.environ.get("SATT_LSTM_CU_CHARGE", "spread")
    flash_bf16 = os.environ.get("SATT_FLASH_BF16", "1") != "0"     # bf16 copies of K | V | Q and d o for the fused attention backward
    head_split = True       # decoder self-attention backward as suffix + prefix launches (backward(): the pipeline starts behind the suffix)
    # low tiles of the split head on the weight-gradient stream beside the loop: MEASURED AND NOT KEPT (8.34 -> 8.43 ms per step, VCTK
    # 5.35 -> 5.44): the head leaves the main stream 90 us earlier, but the fused backward's workgroups (64 KB of LDS, a whole CU's
    # registers) only find CUs in the gaps between two LSTM cluster launches - three gaps of ~25 us every ~240 us - and the chunk
    # that needs their rows then waits ~300 us (attention launch 3.69 -> 3.85 ms).  The switch stays for re-measuring.
    head_split_low = int(os.environ.get("SATT_HEAD_SPLIT_LOW", "0"))       # 0: off, 1: weight-gradient stream, 2: inside the LSTM stream
    head_split_release = int(os.environ.get("SATT_HEAD_SPLIT_RELEASE", "2"))  # chunks before the first one that needs the low rows
    head_split_chunks = int(os.environ.get("SATT_HEAD_SPLIT_CHUNKS", "1"))   # pipeline chunks (from the end) that lie inside the suffix

    def _streams(self):
        if self._side is None:
            self._side = self._device_streams(self.dev)[:2]
        if self.lstm_one_stream:
            return (self._side[0], self._side[0])
        return self._side

    def _t(self, name):
        return Engine._Timed(self, name)

    marks = None    # set to [] to collect (name, event) pairs at the phase boundaries of a step (main stream)

    def _mark(self, name):
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True); ev.record()
            self.marks.append((name, ev))

    side_marks = None    # set to [] to collect (name, event) pairs on whatever stream is current (tools/phase_marks.py: offsets)

    def _side_mark(self, name):
        if self.side_marks is not None:
            ev = torch.cuda.Event(enable_timing=True); ev.record(ops.current_stream())
            self.side_marks.append((name, ev))

    def timing_summary(self):
        """name -> (total ms, launches); call after torch.cuda.synchronize()."""
        return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in (self.timing or {}).items()}

    def _mha_fwd(self, x, prefix, B, T, D, heads, causal, drop, ctx, tag, want_alignments=False, kvq=None):
        """x [B*T, D] -> transformed = x + tanh(Dense(MHA(x)))  (reference modules/module.py:363-371,
        modules/self_attention.py:108-128).  kvq: the K | V | Q projection of x if the caller has made it already (the decoder's
        is produced chunk by chunk inside the recurrent pipeline)."""
        P = self.P
        M, hd = B * T, D // heads
        if kvq is None:
            kvq = self._e(M, 3 * D)
            ops.linear(x, self.W(prefix + ".kvq.W"), P[prefix + ".kvq.b"], kvq)
        nbh = B * heads
        flash = ops.flash_attn_supported(hd) and not want_alignments
        small = not flash and ops.small_attn_supported(hd, T)
        s = p = pd = lse = None
        o = self._e(M, D)
        if small:
            # head depth 16 (the encoder block): QK^T -> softmax -> dropout -> PV in one launch; the probabilities are written
            # once (they are outputs of the block), raw scores and dropped probabilities never exist (csrc/small_attn.hip)
            p = self._e(nbh, T, T)
            ops.small_attn_fwd(kvq, D, p, o, B, T, heads, 1.0 / math.sqrt(hd), drop)
        elif flash:
            # fused QK^T -> causal softmax -> dropout -> PV (csrc/flash.hip): no [B*H, T, T] tensor; backward recomputes P
            lse = self._e(nbh, T)
            # bf16 copies of K | V | Q for the backward kernels, written by the forward kernel as it stages the rows (no launch)
            kvq_b = self._e(M, 3 * D, dtype=torch.bfloat16) if (self.flash_bf16 and ctx.get("training")) else None
            ops.flash_attn_fwd(kvq, D, o, lse, B, T, heads, 1.0 / math.sqrt(hd), causal, drop, kvq_b=kvq_b)
        else:
            s = self._e(nbh, T, T)
            # scores = Q K^T : batch (b outer, head inner)
            ops.gemm(T, T, hd, kvq[:, 2 * D:], 3 * D, kvq, 1, 3 * D, s, T, batch=(B, heads),
                     sA=(T * 3 * D, hd), sB=(T * 3 * D, hd), sC=(heads * T * T, T * T))
            p = self._e(nbh, T, T)
            pd = self._e(nbh, T, T) if drop.thresh else p
            ops.softmax_fwd(s, p, pd if drop.thresh else None, nbh, T, 1.0 / math.sqrt(hd), causal, drop)
            ops.gemm(T, hd, T, pd, T, kvq[:, D:], 3 * D, 1, o, D, batch=(B, heads),
                     sA=(heads * T * T, T * T), sB=(T * 3 * D, hd), sC=(T * D, hd))
        # y = x + tanh((o Wo + bo) Wt + bt) = x + tanh(o Wot + bot): ONE GEMM over the folded pair (_refresh_folded), the
        # residual added after the activation in its epilogue; tanh(.) itself is not kept - the backward pass recovers it as
        # y - x (ops.act_bwd_res) and recomputes o Wo + bo on the weight-gradient stream
        Wd, bd = self._folded[prefix]
        y = self._e(M, D)
        ops.linear(o, Wd, bd.view(-1), y, act=ACT_TANH, residual=x)
        ctx[tag] = dict(x=x, kvq=kvq, p=p, pd=pd, o=o, y=y, s=s, lse=lse, small=small, kvq_b=kvq_b if flash else None)
        return y, p

    def _mha_bwd(self, dy, prefix, B, T, D, heads, causal, drop, c, defer=True, suffix_from=None):
        """returns dx [B*T, D] (gradient wrt the block input, residual path included).
        suffix_from = tA (a multiple of the fused kernel's 64-row tile, causal fused path only): the fused attention backward and
        the K|V|Q input-gradient product run as TWO launches each - rows >= tA of every sample first, then the rows below - and
        self._head_split = (tA, event after the suffix, event after the prefix) lets the caller's backward pipeline (late
        steps first) start on the suffix while the prefix is still running.  The result is accumulated into dy in place."""
        P, G = self.P, self.G
        M, hd = B * T, D // heads
        nbh = B * heads
        kvq, p, pd, o, y, x = c["kvq"], c["p"], c["pd"], c["o"], c["y"], c["x"]
        du = self._e(M, D)
        ops.act_bwd_res(dy, y, x, du, ACT_TANH)

        def tail_dw():      # off the critical path: o2 = o Wo + bo again, d o2 = du Wt^T, then the two weight gradients
            o2, do2 = self._e(M, D), self._e(M, D)
            ops.linear(o, self.W(prefix + ".o.W"), P[prefix + ".o.b"], o2)
            ops.linear_dw(o2, du, G[prefix + ".t.W"], db=G[prefix + ".t.b"])
            ops.linear_dx(du, self.W(prefix + ".t.W"), do2)
            ops.linear_dw(o, do2, G[prefix + ".o.W"], db=G[prefix + ".o.b"])
        self._wgrad(tail_dw, defer=defer)
        do = self._e(M, D)
        ops.linear_dx(du, self._folded[prefix][0], do)          # d o = du (Wo Wt)^T
        dkvq = self._e(M, 3 * D)
        kvq_b = c.get("kvq_b")
        fb = dict(kvq_b=kvq_b, do_b=self._e(M, D, dtype=torch.bfloat16)) if kvq_b is not None else {}
        if c["lse"] is not None and suffix_from and causal:
            # causal: key tile j takes query tiles >= j, query tile i key tiles <= i - ANY tile range leaves its own rows final
            # (include/satt_hip.h satt_flash_attn_bwd_tiles).  Two ranges: the suffix (the pipeline's first chunk waits for it) and
            # the rest (on this stream, in front of the attention kernel).  With head_split_low (OFF: measured slower, see the
            # class attribute) the low half of the rest becomes a third range handed to the caller as a closure, to run beside or
            # inside the loop - its rows are reached more than a millisecond later.
            ts, nt, delta, cur = suffix_from // ops.FLASH_TILE, (T + ops.FLASH_TILE - 1) // ops.FLASH_TILE, self._e(nbh, T), ops.current_stream()
            tm = ts // 2 if (self.head_split_low and self.overlap_wgrad) else 0
            sc, Wk = 1.0 / math.sqrt(hd), self.W(prefix + ".kvq.W")
            ops.flash_attn_bwd(kvq, D, o, do, c["lse"], delta, dkvq, B, T, heads, sc, causal, drop, tiles=(ts, nt), **fb)
            ops.linear_dx_rows(dkvq, Wk, dy, B, T, suffix_from, T, accumulate=True)    # dy = the residual path
            ev_a = torch.cuda.Event(); ev_a.record(cur)
            self._side_mark("head: suffix rows done (main stream)")
            ops.flash_attn_bwd(kvq, D, o, do, c["lse"], delta, dkvq, B, T, heads, sc, causal, drop, tiles=(tm, ts), **fb)
            ops.linear_dx_rows(dkvq, Wk, dy, B, T, tm * ops.FLASH_TILE, suffix_from, accumulate=True)
            ev_b = torch.cuda.Event(); ev_b.record(cur)
            self._side_mark("head: prefix rows done (main stream)")
            kvq_dw = lambda: (ops.linear_dw(x, dkvq, G[prefix + ".kvq.W"], db=G[prefix + ".kvq.b"]))

            def low():          # on the stream the caller chooses, ordered behind ev_b; the caller hands kvq_dw to _wgrad afterwards
                ops.flash_attn_bwd(kvq, D, o, do, c["lse"], delta, dkvq, B, T, heads, sc, causal, drop, tiles=(0, tm), **fb)
                ops.linear_dx_rows(dkvq, Wk, dy, B, T, 0, tm * ops.FLASH_TILE, accumulate=True)
            if tm == 0:
                self._wgrad(kvq_dw, defer=defer)
            self._head_split = (suffix_from, ev_a, ev_b, tm * ops.FLASH_TILE, (low, kvq_dw) if tm else None)
            return dy
        if c["lse"] is not None:          # fused attention: dK | dV | dQ from Q, K, V, o, d o and the saved log-sum-exp
            ops.flash_attn_bwd(kvq, D, o, do, c["lse"], self._e(nbh, T), dkvq, B, T, heads, 1.0 / math.sqrt(hd), causal, drop, **fb)
            self._wgrad(lambda: (ops.linear_dw(x, dkvq, G[prefix + ".kvq.W"], db=G[prefix + ".kvq.b"])), defer=defer)
            dx = self._e(M, D)
            ops.linear_dx(dkvq, self.W(prefix + ".kvq.W"), dx, residual=dy)      # + the residual path's gradient
            return dx
        if c.get("small"):      # fused backward of the head-depth-16 block: two launches (row sums + dQ; dK + dV)
            ops.small_attn_bwd(kvq, D, p, do, dkvq, self._e(nbh, T), B, T, heads, 1.0 / math.sqrt(hd), drop)
            self._wgrad(lambda: (ops.linear_dw(x, dkvq, G[prefix + ".kvq.W"], db=G[prefix + ".kvq.b"])), defer=defer)
            dx = self._e(M, D)
            ops.linear_dx(dkvq, self.W(prefix + ".kvq.W"), dx, residual=dy)
            return dx
        dpd = c["s"]  # reuse the raw-score buffer (not read by any side-stream work)
        # dPd = dO V^T
        ops.gemm(T, T, hd, do, D, kvq[:, D:], 1, 3 * D, dpd, T, batch=(B, heads),
                 sA=(T * D, hd), sB=(T * 3 * D, hd), sC=(heads * T * T, T * T))
        # dV = Pd^T dO
        ops.gemm(T, hd, T, pd, T, do, D, 1, dkvq[:, D:], 3 * D, a_mode=1, batch=(B, heads),
                 sA=(heads * T * T, T * T), sB=(T * D, hd), sC=(T * 3 * D, hd))
        ops.softmax_bwd(dpd, p, dpd, nbh, T, 1.0 / math.sqrt(hd), causal, drop)
        # dQ = dS K ; dK = dS^T Q
        ops.gemm(T, hd, T, dpd, T, kvq, 3 * D, 1, dkvq[:, 2 * D:], 3 * D, batch=(B, heads),
                 sA=(heads * T * T, T * T), sB=(T * 3 * D, hd), sC=(T * 3 * D, hd))
        ops.gemm(T, hd, T, dpd, T, kvq[:, 2 * D:], 3 * D, 1, dkvq, 3 * D, a_mode=1, batch=(B, heads),
                 sA=(heads * T * T, T * T), sB=(T * 3 * D, hd), sC=(T * 3 * D, hd))
        self._wgrad(lambda: (ops.linear_dw(x, dkvq, G[prefix + ".kvq.W"], db=G[prefix + ".kvq.b"])), defer=defer)
        dx = self._e(M, D)
        ops.linear_dx(dkvq, self.W(prefix + ".kvq.W"), dx, residual=dy)
        return dx

    # ------------------------------------------------------------------ forward
    def _encode(self, batch, training, ctx):
        """encoder forward (reference modules/module.py:425-438, :77-110); fills ctx, returns (lstm_out, sa_out)"""
        c, P = self.cfg, self.P
        slen = batch["source_length"]
        # batch["embedded"] ([B, Ti, embedding_dim]): the already embedded inputs, as the reference's encoder layers receive
        # them (modules/module.py:425: call(inputs, input_lengths)) - used by the callable module contracts (modules/module.py)
        B, Ti = batch["embedded"].shape[:2] if "embedded" in batch else batch["source"].shape
        M = B * Ti
        seed = self.seed
        rate = (lambda r: r) if training else (lambda r: 0.0)
        self._wait_shadows()      # bf16 weight shadows refreshed on a side stream after the last update
        # ---- encoder (reference modules/module.py:425-438, :77-110)
        if "embedded" in batch:
            emb = batch["embedded"].to(torch.float32).reshape(M, c.embedding_dim).contiguous()
        else:
            emb = self._e(M, c.embedding_dim)
            ops.embedding_fwd(batch["source"], P["embedding"], emb)
        x = emb
        pre = []
        for n, o in enumerate(c.enc_prenet):
            y = self._e(M, o)
            ops.linear(x, self.W(f"enc.prenet{n}.W"), P[f"enc.prenet{n}.b"], y, act=ACT_RELU,
                       drop=Drop(rate(c.enc_prenet_drop), (S_ENC_PRENET0, S_ENC_PRENET1)[n], seed))
            pre.append(y); x = y
        p1 = x
        CC, K = c.conv_channels, c.max_filter_width
        nb = CC * K
        bank_pre = self._e(M, nb)
        if self._bank_contiguous():
            ops.conv_bank(p1, Ti, self.W("enc.bank1.W"), K, bank_pre)          # all K widths in one launch
        else:
            for k in range(1, K + 1):
                ops.conv1d(p1, Ti, self.W(f"enc.bank{k}.W"), bank_pre[:, (k - 1) * CC:k * CC])
        bn_st = {}

        def bn(xp, name, act):
            Cc = xp.shape[1]
            y = self._e(xp.shape[0], Cc)
            if training:
                mean, rstd = self._e(Cc), self._e(Cc)
                st = self._bn_fused_state(name, xp.shape[0], Cc)       # small activations: one launch (csrc/elementwise.hip)
                if ops.bn_fwd_fused(xp, P[f"enc.{name}.gamma"], P[f"enc.{name}.beta"], y, mean, rstd, self.bn[name][0],
                                    self.bn[name][1], st, c.bn_eps, c.bn_momentum, act):
                    bn_st[name] = (mean, rstd, None)
                else:
                    ws = ops.bn_ws(xp.shape[0], Cc, self.dev)
                    ops.bn_fwd(xp, P[f"enc.{name}.gamma"], P[f"enc.{name}.beta"], y, mean, rstd, self.bn[name][0],
                               self.bn[name][1], ws, c.bn_eps, c.bn_momentum, act)
                    bn_st[name] = (mean, rstd, ws)
            else:
                ops.bn_infer(xp, P[f"enc.{name}.gamma"], P[f"enc.{name}.beta"], self.bn[name][0], self.bn[name][1], y,
                             c.bn_eps, act)
            return y
        # BatchNorm + ReLU + max-pool of the bank in one pass: the activated bank itself is not stored (the backward recomputes it)
        mp = self._e(M, nb)
        bank = None
        if training and bank_pre.is_contiguous():
            mean, rstd = self._e(nb), self._e(nb)
            ws = ops.bn_ws(M, nb, self.dev)
            if ops.bn_maxpool_fwd(bank_pre, P["enc.bank.gamma"], P["enc.bank.beta"], mp, mean, rstd, self.bn["bank"][0],
                                  self.bn["bank"][1], ws, B, Ti, c.bn_eps, c.bn_momentum, ACT_RELU):
                bn_st["bank"] = (mean, rstd, ws)
            else:
                mean = None
        else:
            mean = None
        if mean is None:
            bank = bn(bank_pre, "bank", ACT_RELU)
            ops.maxpool_fwd(bank, mp, B, Ti, nb)
        pr1_pre = self._e(M, c.proj1)
        ops.conv1d(mp, Ti, self.W("enc.proj1.W"), pr1_pre)
        pr1 = bn(pr1_pre, "proj1", ACT_RELU)
        pr2_pre = self._e(M, c.proj2)
        ops.conv1d(pr1, Ti, self.W("enc.proj2.W"), pr2_pre)
        hw = bn(pr2_pre, "proj2", ACT_NONE)
        ops.axpby(p1, hw, 1.0, 1.0)                       # residual (module.py:86)
        H = c.cbhg_out_units // 2
        hws, zs = [hw], []
        hwW = [self.W(f"enc.highway{n}.W") for n in range(c.num_highway)]
        if self.fused_highway and ops.highway_stack_ok(hwW, H):
            # every layer is row-wise: one launch carries 32-row tiles through the whole stack (csrc/highway.hip)
            zs = [self._e(M, 2 * H) for _ in hwW]
            hws += [self._e(M, H) for _ in hwW]
            ops.highway_stack_fwd(hw, hwW, [P[f"enc.highway{n}.b"] for n in range(c.num_highway)], zs, hws[1:])
        else:
            for n in range(c.num_highway):
                z = self._e(M, 2 * H)
                ops.linear(hws[-1], hwW[n], P[f"enc.highway{n}.b"], z)
                y = self._e(M, H)
                ops.highway_fwd(z, hws[-1], y)
                zs.append(z); hws.append(y)
        xg = self._e(2, M, 4 * H)
        for d, nme in enumerate(("fw", "bw")):
            ops.linear(hws[-1], self.W(f"enc.lstm_{nme}.W").rows(0, H), P[f"enc.lstm_{nme}.b"], xg[d])
        lstm_out = self._e(M, 2 * H)
        eg, ecn, ecs, ehs = self._e(2, M, 4 * H), self._e(2, M, H), self._e(2, M, H), self._e(2, M, H)
        self._wait_recurrent_shadows()
        with self._t("enc_lstm_fwd"):
            ops.lstm_fwd(xg, self.shadow["enc.Wh"], slen, 2, B, Ti, H, training, c.zc, c.zh, seed,
                         (S_ENC_FW_C, S_ENC_BW_C), (S_ENC_FW_H, S_ENC_BW_H), lstm_out, eg, ecn, ecs, ehs)
        sa_in = sa_out = enc_align = None
        if c.dual:          # self-attention branch of SelfAttentionCBHGEncoder; ZoneoutEncoderV1 (module.py:336-339) has none
            sa_in = self._e(M, c.sa_units)
            ops.linear(lstm_out, self.W("enc.sa_proj.W"), P["enc.sa_proj.b"], sa_in)
            # self_attention_num_hop stacked blocks with their own weights (modules/module.py:411-419, :433-439)
            sa_out, enc_aligns = sa_in, []
            for h in range(c.sa_num_hop):
                sa_out, al = self._mha_fwd(sa_out, sa_prefix("enc.sa", h), B, Ti, c.sa_units, c.sa_heads, False,
                                           Drop(rate(c.sa_drop), S_ENC_SA + HOP_STREAM * h, seed), ctx, sa_prefix("enc_mha", h),
                                           want_alignments=True)
                enc_aligns.append(al)
            enc_align = enc_aligns[0]
            ctx["enc_aligns"] = enc_aligns
        ctx.update(emb=emb, pre=pre, bank_pre=bank_pre, bank=bank, mp=mp, pr1_pre=pr1_pre, pr1=pr1, pr2_pre=pr2_pre,
                   bn_st=bn_st, hws=hws, zs=zs, enc_lstm=(eg, ecn, ecs, ehs), lstm_out=lstm_out, sa_in=sa_in,
                   sa_out=sa_out, enc_align=enc_align)
        return lstm_out, sa_out

    def forward(self, batch, training=True):
        c, P = self.cfg, self.P
        ctx = {"training": training, "batch": batch}
        slen = batch["source_length"]
        # batch["encoder_outputs"] = (lstm_out [B, Ti, cbhg_out_units], sa_out [B, Ti, sa_units] | None): the decoder half alone on
        # given memories (the reference's decoder call contract, modules/module.py:1493-1498); such a context has no encoder
        # state, so backward() refuses it
        enc_given = batch.get("encoder_outputs")
        B, Ti = enc_given[0].shape[:2] if enc_given is not None else batch["source"].shape
        M = B * Ti
        seed = self.seed
        rate = (lambda r: r) if training else (lambda r: 0.0)
        self._wait_shadows()      # bf16 weight shadows refreshed on a side stream after the last update
        self._mark("fwd start")
        # ---- teacher-input branch of the decoder (reference modules/module.py:1505-1511, helpers.py:42-55): go frame +
        # shifted targets -> pre-net -> input half of the attention-LSTM gates.  It does not depend on the encoder, so
        # it runs on a pipeline stream (idle until the decoder loop) next to the latency-bound encoder forward.
        mel_t = batch["mel"]
        Tm, nm, r = mel_t.shape[1], c.num_mels, c.r
        Td = Tm // r
        Md = B * Td
        feed = nm * c.n_feed_frame
        tg = mel_t.reshape(B, Td, nm * r)
        A = c.att_rnn_units
        G4 = 4 * A
        pn = c.dec_prenet[-1]
        dec_in3 = torch.empty(B, Td, feed, dtype=torch.float32, device=self.dev)
        dec_in = dec_in3.view(Md, feed)
        dpre = [self._e(Md, o) for o in c.dec_prenet]
        xg_att = self._e(Md, G4)
        spk = None
        if c.num_speakers > 0:
            spk = dict(semb=self._e(B, c.speaker_dim), sproj=self._e(B, c.dec_prenet[0]),
                       r0=self._e(Md, c.dec_prenet[0]), d0=self._e(Md, c.dec_prenet[0]))

        def teacher_branch():
            dec_in3[:, :1].zero_()                                              # go frame
            dec_in3[:, 1:] = tg[:, :-1, nm * r - feed:]                         # shifted targets
            if spk is not None:
                # MultiSpeakerPreNet (reference modules/multi_speaker_modules.py:27-32; models/models.py:298-301,
                # 338-339): dense0 = relu(x W0 + b0) + softsign(emb[speaker] Ws + bs); dense = relu(dense0 W2 + b2)
                if "speaker_embed" in batch:      # the embedded speaker vectors themselves (decoder call contract: speaker_embed=)
                    spk["semb"].copy_(batch["speaker_embed"].reshape(B, c.speaker_dim))
                else:
                    ops.embedding_fwd(batch["speaker_id"], P["speaker_embedding"], spk["semb"], offset=c.speaker_offset)
                ops.linear(spk["semb"], self.W("dec.prenet0.Ws"), P["dec.prenet0.bs"], spk["sproj"], act=ACT_SOFTSIGN)
                ops.linear(dec_in, self.W("dec.prenet0.W"), P["dec.prenet0.b"], spk["r0"], act=ACT_RELU)
                ops.axpby(spk["r0"], spk["d0"], 1.0, 0.0)
                ops.bcast_add(spk["sproj"], spk["d0"], B, Td, c.dec_prenet[0])
            x = dec_in
            for n, o in enumerate(c.dec_prenet):
                y = dpre[n]
                if n == 0 and spk is not None:
                    ops.linear(spk["d0"], self.W("dec.prenet0.W2"), P["dec.prenet0.b2"], y, act=ACT_RELU,
                               drop=Drop(rate(c.dec_prenet_drop), S_DEC_PRENET0, seed))
                else:       # the plain PreNet layers keep their dropout outside training with apply_dropout_on_inference
                    ops.linear(x, self.W(f"dec.prenet{n}.W"), P[f"dec.prenet{n}.b"], y, act=ACT_RELU,
                               drop=Drop(self.dec_prenet_rate(training, plain=True), (S_DEC_PRENET0, S_DEC_PRENET1)[n], seed))
                x = y
            ops.linear(dpre[-1], self.W("dec.att_lstm.W").rows(0, pn), P["dec.att_lstm.b"], xg_att)

        main0 = ops.current_stream()
        side = self._streams()[0] if self.overlap_wgrad else main0
        if side is not main0:
            ev_in = torch.cuda.Event(); ev_in.record(main0)
            side.wait_event(ev_in)
            with ops.on_stream(side):
                teacher_branch()
                ev_teacher = torch.cuda.Event(); ev_teacher.record(side)
        if enc_given is not None:
            lstm_out = enc_given[0].to(torch.float32).reshape(M, c.cbhg_out_units).contiguous()
            sa_out = enc_given[1].to(torch.float32).reshape(M, c.sa_units).contiguous() if c.dual else None
            ctx.update(lstm_out=lstm_out, sa_out=sa_out, enc_align=None, decoder_only=True)
        else:
            lstm_out, sa_out = self._encode(batch, training, ctx)
        # the loss denominators depend on the batch only: summed here (the encoder is enqueued: none of this host work sits in
        # front of the step's first kernels) on the weight-gradient stream, so that the loss is ONE launch between the forward
        # and the backward pass (ops.loss_fwd_bwd_presummed).  The chunk counters of the two
        # single-launch attention kernels come from a two-block ring: this step uses the block that was zeroed during the previous
        # step, and zeroes the other one here, off the critical path (a torch.zeros in front of each kernel was a dependent launch)
        Bm, Tmm = batch["mel"].shape[0], batch["mel"].shape[1]
        if self._ctr is None:
            self._ctr = torch.zeros(2, 48, dtype=torch.int32, device=self.dev)
        self._ctr_par ^= 1
        ctr_next = self._ctr[self._ctr_par ^ 1]
        if self.overlap_wgrad:
            if self._wg_stream is None:
                self._wg_stream = self._device_streams(self.dev)[2]
            if side is not main0:
                self._wg_stream.wait_event(ev_in)       # the batch was uploaded before the start of forward(): no new record
            else:
                ev = torch.cuda.Event(); ev.record(ops.current_stream())
                self._wg_stream.wait_event(ev)
            with ops.on_stream(self._wg_stream):
                ops.loss_mask_sums(batch["spec_loss_mask"], batch["binary_loss_mask"], Bm, Tmm, Tmm // c.r, self._loss_ws)
                ctr_next.zero_()
                self._loss_ev = torch.cuda.Event(); self._loss_ev.record(self._wg_stream)
        else:
            ops.loss_mask_sums(batch["spec_loss_mask"], batch["binary_loss_mask"], Bm, Tmm, Tmm // c.r, self._loss_ws)
            ctr_next.zero_()
            self._loss_ev = None
        if self._ctr_ev is not None:          # this step's block: zeroed on the weight-gradient stream one step ago
            ops.current_stream().wait_event(self._ctr_ev)
        self._ctr_ev = self._loss_ev
        if side is main0:
            teacher_branch()
        else:
            main0.wait_event(ev_teacher)
        self._mark("encoder fwd")
        ctx["spk"] = spk
        V1, V2, U1, U2 = c.cbhg_out_units, c.sa_units, c.att1_units, c.att2_units
        CT = V1 + V2
        # single source (c.dual False: V2 = U2 = 0, AttentionRNN of ExtendedDecoder, reference modules/module.py:566-574):
        # the second mechanism's pointers stay NULL - the kernels then see zero energies and an empty second context
        # benchmark precision: the attention memory is held at bf16 precision (in fp32 storage) - the attention kernels, the
        # key projection and the memory gradients all read it as bf16 anyway; the backward loop relies on both passes seeing
        # the SAME value rows (its sums over all memory rows come from the saved contexts, csrc/attn_cluster.hip phase (b))
        mem_bf16 = ops.get_precision() == "bf16"
        values1 = self._e(M, V1)
        ops.seq_mask(lstm_out, slen, values1, B, Ti, V1, round_bf16=mem_bf16)
        keys1 = self._e(M, U1)
        ops.linear(values1, self.W("dec.att1.Wm"), None, keys1)
        values2 = keys2 = None
        if c.dual:
            values2, keys2 = self._e(M, V2), self._e(M, U2)
            ops.seq_mask(sa_out, slen, values2, B, Ti, V2, round_bf16=mem_bf16)
            ops.linear(values2, self.W("dec.att2.Wm"), None, keys2)
        att_out = self._e(Md, A + CT)
        al1, al2, a1 = self._e(B, Td, Ti), self._e(B, Td, Ti), self._e(B, Td, Ti)
        pq = self._e(Md, U1 + U2)
        flb = self._e(Md * Ti, c.att_filters)
        ag, acn, acs, ahs = self._e(Md, G4), self._e(Md, A), self._e(Md, A), self._e(Md, A)
        zct, _ = ops.rate_thresh(c.zc if training else 0.0)
        zht, _ = ops.rate_thresh(c.zh if training else 0.0)
        ap = ops.attn_rnn_params(
            B=B, Td=Td, Ti=Ti, A=A, U1=U1, V1=V1, U2=U2, V2=V2, kernel=c.att_kernel, filters=c.att_filters,
            training=int(training), keys_lds_bf16=int(ops.get_precision() == "bf16"), zc=c.zc, zh=c.zh, zc_thresh=zct, zh_thresh=zht, seed=seed,
            stream_c=S_ATT_C, stream_h=S_ATT_H, lengths=slen, xg=xg_att, Wrec=self.shadow["att.Wrec"],
            Wq=self.shadow["att.Wq"], keys1=keys1, values1=values1, keys2=keys2, values2=values2,
            locF=P["dec.att1.F"], locFb=P["dec.att1.bF"], locU=P["dec.att1.U"], v1=P["dec.att1.v"],
            b1=P["dec.att1.b"], v2=P.get("dec.att2.v"), out=att_out, align1=al1, align2=al2, a1=a1, pq=pq,
            fl=flb, gates=ag, cnew=acn, cstate=acs, hstate=ahs,
            att1_mode=int(c.attention == "location_sensitive"), cumulative=int(c.cumulative_weights))
        ap_acum = None
        if c.cumulative_weights:
            ap_acum = self._e(B, Td, Ti)
            ap.acum = ap_acum.data_ptr()
        ustate = None
        if c.transition_agent:      # u of the forward recursion predicted per step (modules/forward_attention.py:111-116)
            ustate = self._e(B, Td)
            ap.agentW, ap.agentb, ap.ustate = P["dec.att1.Wa"].data_ptr(), P["dec.att1.ba"].data_ptr(), ustate.data_ptr()
        Ca = ops.attn_cluster_size(ap) if self.use_clusters else 0
        if not Ca and (c.attention != "forward" or c.cumulative_weights or c.transition_agent):
            from .modules.attentions import UnsupportedConfiguration
            raise UnsupportedConfiguration("attention=%s cumulative_weights=%s transition_agent=%s needs the cluster attention "
                                           "kernels, which do not accept this problem (B=%d, Ti=%d)"
                                           % (c.attention, c.cumulative_weights, c.transition_agent, B, Ti))
        D = c.dec_units
        Cn = ops.lstm_cluster_size(B, D) if self.use_clusters else 0
        aws = None
        vw1, fold_pack, ctx1_rows = None, None, None
        if Ca:
            if Ca not in self._pack_cache:
                self._pack_cache[Ca] = ops.attn_cluster_pack(P["dec.att_lstm.W"][pn:], A, Ca)
            aws = self._cluster_ws("attn", lambda: ops.attn_cluster_ws(ap, Ca, self.dev), (B, Ti, Ca, A, U1, U2, V1, V2))
            if self.fold_context and ops.attn_cluster_fold(ap, Ca):
                # FOLDED first-source context (csrc/attn_cluster.hip, FOLD): gates += ctx1 Wc1 = alpha (values1 Wc1).  VW1 is one
                # GEMM per step; the forward kernel then multiplies the alignments with its own columns of it inside the recurrent
                # product and never forms ctx1 - which remains an OUTPUT (LSTM1's input, the backward pass): alpha x values1 per
                # pipeline chunk on the LSTM1 stream, exact fp32 (the backward kernel's row sums rely on ctx1 = sum alpha v)
                key = ("fold", Ca)
                if key not in self._pack_cache:
                    self._pack_cache[key] = ops.attn_cluster_pack(P["dec.att_lstm.W"][pn + V1:], A, Ca)
                fold_pack = self._pack_cache[key][0]
                vw1 = self._e(M, G4)
                ops.linear(values1, self.W("dec.att_lstm.W").rows(pn, pn + V1), None, vw1)
                al1_rows = al1.view(Md, Ti)
                if training and self.save_attention_factors:
                    # the folded forward kernel also leaves the derivative factors r (1 - r) of the energy nonlinearity (fp16,
                    # [B, Td, Ti, U1 + U2]: 1 GB at the benchmark shape - HBM is what this chip has plenty of); the backward
                    # kernel reads them instead of recomputing keys + query + location term -> exp2 -> rcp per (row, unit)
                    saf = self._e(Md * Ti * (U1 + U2), dtype=torch.float16)
                    ap.saf = saf.data_ptr()
                    ctx["saf"] = saf

                def ctx1_rows(t0, t1):
                    ops.gemm(t1 - t0, V1, Ti, al1_rows[t0:], Ti, values1, V1, 1, att_out[t0:, A:], A + CT, batch=(B, 1),
                             sA=(Td * Ti, 0), sB=(Ti * V1, 0), sC=(Td * (A + CT), 0), prec=ops.PREC_F32)
        lp1 = lp2 = lin = None
        if Cn:
            lp1, lp2 = self.lstm_cluster_packs(Cn)
            if self.fuse_xg_steps > 0 and ops.get_precision() == "bf16":
                lin = self.lstm_cluster_in_packs(Cn)
        xg1, xg2 = self._e(1, Md, 4 * D), self._e(1, Md, 4 * D)
        h1, dec_out = self._e(Md, D), self._e(Md, D)
        l1 = (self._e(1, Md, 4 * D), self._e(1, Md, D), self._e(1, Md, D), self._e(1, Md, D))
        l2 = (self._e(1, Md, 4 * D), self._e(1, Md, D), self._e(1, Md, D), self._e(1, Md, D))
        # the decoder self-attention's K | V | Q projection is a row-wise product of the LSTM2 output: made per pipeline chunk on the
        # LSTM2 stream, right behind the chunk, instead of for all rows after the loop (one 23 us GEMM off the chain between the loops)
        kvq_dec = self._e(Md, 3 * c.dec_sa_units) if (c.dec_sa_units > 0 and self.pipeline_kvq) else None
        cws1 = self._cluster_ws("lstm1", lambda: ops.lstm_cluster_ws(B, D, Cn, self.dev), (B, D, Cn)) if Cn else None
        cws2 = self._cluster_ws("lstm2", lambda: ops.lstm_cluster_ws(B, D, Cn, self.dev), (B, D, Cn)) if Cn else None
        NC = max(1, min(self.pipeline_chunks, Td)) if (Ca and Cn) else 1
        if NC > 1 and not self._layers_fit_side_by_side(ap, Ca, vw1, B, Td, D, Cn):
            # The layer pipeline keeps the attention kernel (one workgroup per CU, the whole register file) and ONE LSTM cluster
            # kernel in flight together, and every member of a cluster spins for its peers: unless every workgroup of both can
            # have a CU of its own (B <= 32 with clusters of 4 on 256 CUs) they can end up partially resident, each waiting for
            # workgroups the other one keeps out - measured at B = 48 / 64 as 3 s of hand-off timeouts per step, and at B = 40 / 42
            # even when two LSTM workgroups per CU would make the sum fit (B = 33, 36 happened to run).  Larger batches run the
            # three layers one after the other (B = 64: 13.5 ms per step, 3.8 M frames/s).  r5: the footprints come from the
            # occupancy calculator for the kernels that will actually be launched (_layers_fit_side_by_side), not from an assumed
            # one-workgroup-per-CU device of 256 CUs.
            NC = 1
        if NC > 1:
            # The three recurrent layers form a producer/consumer chain and each cluster kernel occupies only
            # B*C CUs: run them as a software pipeline over time chunks on three HIP streams.
            main = ops.current_stream()
            s1, s2 = self._streams()
            bounds = self._chunk_bounds(Td, NC, self.pipeline_tail_fwd)
            ev1 = None
            # ONE attention launch over all steps (one prologue instead of one per chunk): the kernel counts its finished
            # chunks in `prog` and the LSTM1 stream waits on the counter (hipStreamWaitValue32) instead of on kernel ends
            # (not when kernels of different streams cannot overlap - counter-collecting profilers serialise them, and a
            # serialised wait would sit in front of the kernel it waits for)
            single = self.single_launch_attention and len(bounds) <= 16 and ops.streams_run_concurrently(main, s1)
            if single:
                prog = self._ctr[self._ctr_par][0:16]
                self._keep_fwd = prog
                evz = torch.cuda.Event(); evz.record(main)          # the zeroed counter, before the kernel starts
                with self._t("attn_rnn_fwd"):
                    ops.attn_cluster_fwd(ap, Ca, fold_pack if vw1 is not None else self._pack_cache[Ca][0], aws, 0, Td,
                                         progress=prog, bounds=[b1 for (_, b1) in bounds], vw1=vw1)
            for k, (t0, t1) in enumerate(bounds):
                if not single:
                    with self._t("attn_rnn_fwd"):
                        ops.attn_cluster_fwd(ap, Ca, fold_pack if vw1 is not None else self._pack_cache[Ca][0], aws, t0, t1, vw1=vw1)
                    eva = torch.cuda.Event(); eva.record(main)
                with ops.on_stream(s1):
                    if single:
                        if k == 0:
                            s1.wait_event(evz)
                        ops.stream_wait_value(prog[k:k + 1], B * Ca, s1)       # every workgroup has finished chunk k
                    else:
                        s1.wait_event(eva)
                    if ctx1_rows is not None:
                        ctx1_rows(t0, t1)
                    fuse = lin is not None and t1 - t0 <= self.fuse_xg_steps
                    if fuse and lin[0] is not None:
                        with self._t("lstm1_fwd"):
                            ops.lstm_cluster_fwd_x(att_out, A + CT, lin[0], P["dec.lstm1.b"], xg1, lp1[0], B, Td, D, Cn, training, c.zc,
                                                   c.zh, seed, S_L1_C, S_L1_H, h1, l1[0], l1[1], l1[2], l1[3], cws1, t0, t1)
                    else:
                        ops.linear_rows(att_out, self.W("dec.lstm1.W").rows(0, A + CT), P["dec.lstm1.b"], xg1[0], B, Td, t0, t1)
                        with self._t("lstm1_fwd"):
                            ops.lstm_cluster_fwd(xg1, lp1[0], B, Td, D, Cn, training, c.zc, c.zh, seed,
                                                 S_L1_C, S_L1_H, h1, l1[0], l1[1], l1[2], l1[3], cws1, t0, t1)
                    if s2 is not s1:       # (one LSTM stream: program order; an event pair is two marker packets in the chain)
                        ev1 = torch.cuda.Event(); ev1.record(s1)
                with ops.on_stream(s2):
                    if s2 is not s1:
                        s2.wait_event(ev1)
                    if fuse and lin[1] is not None:
                        with self._t("lstm2_fwd"):
                            ops.lstm_cluster_fwd_x(h1, D, lin[1], P["dec.lstm2.b"], xg2, lp2[0], B, Td, D, Cn, training, c.zc, c.zh,
                                                   seed, S_L2_C, S_L2_H, dec_out, l2[0], l2[1], l2[2], l2[3], cws2, t0, t1)
                    else:
                        ops.linear_rows(h1, self.W("dec.lstm2.W").rows(0, D), P["dec.lstm2.b"], xg2[0], B, Td, t0, t1)
                        with self._t("lstm2_fwd"):
                            ops.lstm_cluster_fwd(xg2, lp2[0], B, Td, D, Cn, training, c.zc, c.zh, seed,
                                                 S_L2_C, S_L2_H, dec_out, l2[0], l2[1], l2[2], l2[3], cws2, t0, t1)
                    if kvq_dec is not None and self.pipeline_kvq:
                        ops.linear_rows(dec_out, self.W("dec.sa.kvq.W"), P["dec.sa.kvq.b"], kvq_dec, B, Td, t0, t1)
            kvq_done = kvq_dec is not None and self.pipeline_kvq
            ev2 = torch.cuda.Event(); ev2.record(s2)
            main.wait_event(ev2)
        else:
            with self._t("attn_rnn_fwd"):
                if Ca:
                    ops.attn_cluster_fwd(ap, Ca, fold_pack if vw1 is not None else self._pack_cache[Ca][0], aws, vw1=vw1)
                else:
                    ops.attn_rnn_fwd(ap)
            if ctx1_rows is not None:
                ctx1_rows(0, Td)
            ops.linear(att_out, self.W("dec.lstm1.W").rows(0, A + CT), P["dec.lstm1.b"], xg1[0])
            with self._t("lstm1_fwd"):
                if Cn:
                    ops.lstm_cluster_fwd(xg1, lp1[0], B, Td, D, Cn, training, c.zc, c.zh, seed, S_L1_C,
                                         S_L1_H, h1, l1[0], l1[1], l1[2], l1[3], cws1)
                else:
                    ops.lstm_fwd(xg1, self.shadow["l1.Wh"], None, 1, B, Td, D, training, c.zc, c.zh, seed, (S_L1_C,),
                                 (S_L1_H,), h1, *l1)
            ops.linear(h1, self.W("dec.lstm2.W").rows(0, D), P["dec.lstm2.b"], xg2[0])
            with self._t("lstm2_fwd"):
                if Cn:
                    ops.lstm_cluster_fwd(xg2, lp2[0], B, Td, D, Cn, training, c.zc, c.zh, seed, S_L2_C,
                                         S_L2_H, dec_out, l2[0], l2[1], l2[2], l2[3], cws2)
                else:
                    ops.lstm_fwd(xg2, self.shadow["l2.Wh"], None, 1, B, Td, D, training, c.zc, c.zh, seed, (S_L2_C,),
                                 (S_L2_H,), dec_out, *l2)
        if NC <= 1:
            kvq_done = False
        ctx["att_cluster"] = (Ca, aws)
        ctx["cluster"] = (Cn, cws1, cws2)
        ctx["chunks"] = NC
        ctx["single_launch_fwd"] = bool(NC > 1 and single)       # which attention schedule ran (tests assert it)
        self._mark("decoder loop fwd")
        if c.dec_sa_units > 0:
            # decoder_self_attention_num_hop stacked causal blocks (modules/module.py:707-715, :753-757); the first hop's
            # K | V | Q projection was made chunk by chunk inside the recurrent pipeline
            tr = dec_out
            for h in range(c.dec_sa_num_hop):
                tr, dec_align = self._mha_fwd(tr, sa_prefix("dec.sa", h), B, Td, c.dec_sa_units, c.dec_sa_heads, True,
                                              Drop(rate(c.dec_sa_drop), S_DEC_SA + HOP_STREAM * h, seed), ctx,
                                              sa_prefix("dec_mha", h), kvq=kvq_dec if (kvq_done and h == 0) else None)
        else:               # ExtendedDecoder: OutputAndStopTokenWrapper projects the DecoderRNNV2 output (module.py:588-590)
            tr = dec_out
        NO = nm * r + 1
        NOp = NO + self._out_pad()      # rows padded to whole 8-column groups (16-byte rows for the input-gradient GEMM's operands)
        yout = self._e(Md, NOp)[:, :NO]                     # [mel frames of the step | stop logit]
        ops.linear(tr, self.W("dec.out.W"), P["dec.out.b"], yout)
        ctx.update(dec_in=dec_in, dpre=dpre, values1=values1, values2=values2, keys1=keys1, keys2=keys2,
                   att_params=ap, att_out=att_out, al1=al1, al2=al2, a1=a1, pq=pq, flb=flb, acum=ap_acum, ustate=ustate,
                   att_saved=(ag, acn, acs, ahs),
                   h1=h1, l1=l1, l2=l2, dec_out=dec_out, tr=tr, yout=yout, dims=(B, Ti, Td, Tm))
        self._mark("decoder head fwd")
        # ---- losses (+ gradient wrt yout)
        dyp = self._e(Md, NOp)          # the loss kernel zero-fills the pad columns behind [d mel | d stop]
        dy = dyp[:, :NO]
        if self._loss_ev is not None:
            ops.current_stream().wait_event(self._loss_ev)
        ops.loss_fwd_bwd_presummed(yout, NOp, mel_t, batch["spec_loss_mask"], yout[:, NO - 1:], NOp, batch["done"],
                                   batch["binary_loss_mask"], B, Tm, nm, Td, self.loss_l2, self.losses, dy, NOp,
                                   dy[:, NO - 1:], NOp, self._loss_ws)
        ctx["dy"], ctx["dy_padded"] = dy, dyp
        if self._l2_n and training:       # + scale * sum ||W||^2 / 2; its gradient goes straight into the flat gradient buffer
            self.reg_loss.zero_()
            ops.l2_reg(self.flat, self.grad, self._l2_table, self._l2_n, c.l2_weight, self.reg_loss, self.losses[2:3])
        self._mark("loss")
        if c.use_postnet_v2:
            self._postnet(ctx, yout, dy, batch, training)
        return ctx

    def _postnet(self, ctx, yout, dy, batch, training):
        """PostNetV2 (reference models/models.py:92-100,116-118): num_layers x [Conv1d(k) -> BN -> tanh
        (last: linear) -> dropout], Dense(C -> num_mels), residual; extra spec_loss term.  The loss gradient is born in
        forward() (fused loss kernel), so this block runs its own backward right away: parameter gradients accumulate
        into self.grad and dL/dmel is added to ctx['dy'] before Engine.backward() consumes it."""
        c, P, G = self.cfg, self.P, self.G
        B, Ti, Td, Tm = ctx["dims"]
        nm, r = c.num_mels, c.r
        NO, W = nm * r + 1, nm * r
        Md, Mm, Co, L = B * Td, B * Tm, c.postnet_v2_out_channels, c.num_postnet_v2_layers
        keep = ctx.setdefault("_postnet_keep", [])
        mel_c = self._e(Md, W)
        ops.axpby(yout[:, :W], mel_c, 1.0, 0.0)
        x = mel_c.view(Mm, nm)
        saved = []
        for n in range(L):
            pre = self._e(Mm, Co)
            ops.conv1d(x, Tm, self.W(f"postnet.conv{n}.W"), pre)
            act = ACT_TANH if n < L - 1 else ACT_NONE
            y = self._e(Mm, Co)
            name, st = f"postnet{n}", None
            if training:
                mean, rstd = self._e(Co), self._e(Co)
                ws = ops.bn_ws(Mm, Co, self.dev)
                ops.bn_fwd(pre, P[f"postnet.bn{n}.gamma"], P[f"postnet.bn{n}.beta"], y, mean, rstd, self.bn[name][0],
                           self.bn[name][1], ws, c.bn_eps, c.bn_momentum, act)
                st = (mean, rstd, ws)
            else:
                ops.bn_infer(pre, P[f"postnet.bn{n}.gamma"], P[f"postnet.bn{n}.beta"], self.bn[name][0], self.bn[name][1],
                             y, c.bn_eps, act)
            drop = Drop(c.postnet_v2_drop_rate if training else 0.0, S_POSTNET0 + n, self.seed)
            d = self._e(Mm, Co)
            ops.dropout(y, d, drop)
            saved.append((x, pre, st, drop, act))
            keep += [x, pre, y, d]
            x = d
        proj = self._e(Mm, nm)
        ops.linear(x, self.W("postnet.proj.W"), P["postnet.proj.b"], proj)
        post = self._e(Md, W)
        ops.axpby(mel_c, post, 1.0, 0.0)
        ops.axpby(proj.view(Md, W), post, 1.0, 1.0)
        dpost = self._e(Md, W)
        ops.loss_fwd_bwd(post, W, batch["mel"], batch["spec_loss_mask"], yout[:, NO - 1:], yout.stride(0), batch["done"],
                         batch["binary_loss_mask"], B, Tm, nm, Td, self.loss_l2, self.post_losses, dpost, W, None, 0,
                         self._loss_ws2)
        self.losses[2:3].add_(self.post_losses[0:1])             # loss = mel + done + postnet_mel (models.py:118)
        ctx["mel_postnet"] = post
        keep += [mel_c, proj, post, dpost]
        if not training:
            return
        dproj = dpost.view(Mm, nm)
        xl = x
        self._wgrad(lambda: (ops.linear_dw(xl, dproj, G["postnet.proj.W"], db=G["postnet.proj.b"])))
        dx = self._e(Mm, Co)
        ops.linear_dx(dproj, self.W("postnet.proj.W"), dx)
        for n in reversed(range(L)):
            xin, pre, st, drop, act = saved[n]
            dyv = self._e(Mm, Co)
            ops.dropout(dx, dyv, drop)
            dpre = self._e(Mm, Co)
            ops.bn_bwd(dyv, pre, P[f"postnet.bn{n}.gamma"], P[f"postnet.bn{n}.beta"], st[0], st[1], dpre,
                       G[f"postnet.bn{n}.gamma"], G[f"postnet.bn{n}.beta"], st[2], act)
            self._wgrad(lambda: ops.conv1d_dw(xin, Tm, dpre, G[f"postnet.conv{n}.W"]))
            keep += [dx, dyv, dpre]
            dx = self._e(Mm, xin.shape[1])
            ops.conv1d_dx(dpre, Tm, self.W(f"postnet.conv{n}.W"), dx)
        keep.append(dx)
        ops.axpby(dpost, dy[:, :W], 1.0, 1.0)                      # residual path
        ops.axpby(dx.view(Md, W), dy[:, :W], 1.0, 1.0)             # through the conv stack

    _ws_cache = None
    _ws_last = None

    def _cluster_ws(self, kind, make, key):
        """Exchange workspace of a cluster kernel family, allocated (zero-filled) ONCE per problem shape and kept: launches
        clear the granules only, so the error word in the tail is sticky across steps - a hand-off timeout of any step is
        seen by the optimiser kernel of that step (update skipped) and by the next check_clusters() however rarely the host
        looks.  Shapes alternate (train / eval batches): one workspace per (family, shape)."""
        if self._ws_cache is None:
            self._ws_cache, self._ws_last = {}, {}
        ws = self._ws_cache.get((kind, key))
        if ws is None:
            if len(self._ws_cache) >= 24:       # many distinct batch shapes (unbucketed data): forget the oldest
                self._ws_cache.pop(next(iter(self._ws_cache)))
            ws = self._ws_cache[(kind, key)] = make()
        self._ws_last[kind] = ws                # the workspaces of the latest step: what optimizer_step guards on
        return ws

    def recover_from_handoff_timeout(self):
        """After check_clusters() raised: fall back to the schedule without in-kernel chunk hand-offs (one attention launch per
        pipeline chunk, every launch ordered by stream events), clear the sticky error words and carry on.  The updates of the
        affected steps were skipped on the device (satt_adam_step's err arguments), so the parameters are intact.  Returns False
        if the fallback schedule was already active (then the timeout has another cause: the caller should stop)."""
        if not self.single_launch_attention:
            return False
        self.single_launch_attention = False
        torch.cuda.synchronize()
        for ws in (self._ws_cache or {}).values():
            ws[-64:].zero_()
        return True

    def check_clusters(self, ctx):
        """host-synchronous: raise if any inter-workgroup hand-off of the cluster kernels timed out - in this step or in any
        earlier one since the workspaces were allocated (the error words are sticky, see _cluster_ws)"""
        Ca, aws = ctx.get("att_cluster", (0, None))
        if Ca:
            ops.attn_cluster_status(ctx["att_params"], Ca, aws)
        Cn, cws1, cws2 = ctx.get("cluster", (0, None, None))
        if Cn:
            B, _, Td, _ = ctx["dims"]
            ops.lstm_cluster_status(cws1, B, self.cfg.dec_units, Cn)
            ops.lstm_cluster_status(cws2, B, self.cfg.dec_units, Cn)

    def outputs(self, ctx):
        """Views of the step's results in the reference's layouts (models/models.py:397-408)."""
        B, Ti, Td, Tm = ctx["dims"]
        c = self.cfg
        y = ctx["yout"]
        mel = y[:, :-1].reshape(B, Td * c.r, c.num_mels)
        out = dict(mel=mel, stop=y[:, -1:].reshape(B, Td, 1), alignment1=ctx["al1"], lstm_out=ctx["lstm_out"].view(B, Ti, -1),
                   dec_out=ctx["dec_out"].view(B, Td, -1), mel_loss=self.losses[0], done_loss=self.losses[1],
                   loss=self.losses[2])
        if c.dual:       # second attention history + encoder self-attention heads (models/models.py:397-408)
            out.update(alignment2=ctx["al2"], enc_alignment=None if ctx["enc_align"] is None else ctx["enc_align"].view(B, c.sa_heads, Ti, Ti),
                       sa_out=ctx["sa_out"].view(B, Ti, -1))
            if c.sa_num_hop > 1 and "enc_aligns" in ctx:       # the reference collects the alignments of every hop (modules/module.py:433-439)
                out.update(enc_alignments=torch.stack([a.view(B, c.sa_heads, Ti, Ti) for a in ctx["enc_aligns"]]))   # [hop, B, heads, Ti, Ti]
        if c.use_postnet_v2:
            out.update(mel_postnet=ctx["mel_postnet"].view(B, Tm, c.num_mels), postnet_mel_loss=self.post_losses[0])
        if self._l2_n:
            out.update(regularization_loss=self.reg_loss[0])
        return out

    # ------------------------------------------------------------------ backward
    def _fire_behind_wgrads(self, callback):
        """run `callback` (a gradient-bucket all-reduce) ordered after every weight-gradient launch issued so far, on the
        weight-gradient stream: the main stream's backward chain is not blocked"""
        if self.overlap_wgrad:
            if self._wg_stream is None:
                self._wg_stream = self._device_streams(self.dev)[2]
            ev = torch.cuda.Event(); ev.record(ops.current_stream())
            self._wg_stream.wait_event(ev)
            for e in (self._join or ()):
                self._wg_stream.wait_event(e)
            self._wgrad_gather(self._wg_stream)
            with ops.on_stream(self._wg_stream):
                callback()
            if self._wg_used is None:
                self._wg_used = []
            if self._wg_stream not in self._wg_used:
                self._wg_used.append(self._wg_stream)
        else:
            for e in (self._join or ()):
                ops.current_stream().wait_event(e)
            self._join = None
            callback()

    def backward(self, ctx, on_decoder_grads_ready=None, on_upper_encoder_grads_ready=None):
        """Hand-written backward of forward(); parameter gradients are ACCUMULATED into self.grad (zero it first).
        `on_decoder_grads_ready` fires once every decoder-parameter gradient is final (DP bucket 1);
        `on_upper_encoder_grads_ready` once the encoder's gradients from enc.proj1.W upwards (projections, highway, BiLSTM,
        self-attention: ~5 MB) are - in front of the conv-bank backward, whose 9.6 MB of gradients close the step (DP bucket 2)."""
        c, P, G = self.cfg, self.P, self.G
        if ctx.get("decoder_only"):
            raise ops._lib.SattError("backward: this forward ran the decoder half on given encoder outputs (no encoder state to "
                                "differentiate through); train through Engine.train_step")
        training = ctx["training"]
        B, Ti, Td, Tm = ctx["dims"]
        M, Md = B * Ti, B * Td
        seed = self.seed
        self._keep = []     # temporaries are read by side streams: keep them allocated until the final join
        rate = (lambda r: r) if training else (lambda r: 0.0)
        slen = ctx["batch"]["source_length"]
        dy, tr = ctx["dy"], ctx["tr"]
        D, A = c.dec_units, c.att_rnn_units
        V1, V2, U1, U2 = c.cbhg_out_units, c.sa_units, c.att1_units, c.att2_units
        CT = V1 + V2
        # ---- output projection
        self._wgrad(lambda: (ops.linear_dw(tr, dy, G["dec.out.W"], db=G["dec.out.b"])))
        dtr = self._e(Md, c.out_in)
        dyp, Wn = ctx["dy_padded"], self.shadow.get("out.Wn")
        # K = 161 has no 16-byte rows: the padded gradient rows against the zero-padded bf16 shadow run on the large-tile kernel
        if not (Wn is not None and dyp.shape[1] == Wn.shape[1] and ops.gemm(
                Md, c.out_in, dyp.shape[1], dyp, dyp.shape[1], P["dec.out.W"], 1, dy.shape[1], dtr, c.out_in, Bs=Wn,
                sbs_n=Wn.shape[1], only_path=1)):
            ops.linear_dx(dy, self.W("dec.out.W"), dtr)
        ddec = dtr
        self._head_split = None
        if c.dec_sa_units > 0:
            ddec = dtr
            # The backward pipeline below walks the decoder steps late to early and its first chunk is short: under the causal mask
            # the gradient rows of the LAST chunk(s) only need the suffix tiles of the fused attention backward, so that launch is
            # split at the 64-row tile boundary at or below the last chunk's start and the pipeline starts behind the suffix.
            t_a = 0
            if self.head_split and c.dec_sa_num_hop == 1 and ctx["chunks"] > 1:
                t_a = self._head_split_row(self._chunk_bounds(Td, ctx["chunks"]), self.head_split_chunks, ops.FLASH_TILE)
            for h in reversed(range(c.dec_sa_num_hop)):
                ddec = self._mha_bwd(ddec, sa_prefix("dec.sa", h), B, Td, c.dec_sa_units, c.dec_sa_heads, True,
                                     Drop(rate(c.dec_sa_drop), S_DEC_SA + HOP_STREAM * h, seed), ctx[sa_prefix("dec_mha", h)],
                                     defer=False, suffix_from=t_a)
        # (the head's weight gradients are NOT deferred: launched at the end of the head they were still running when the
        # recurrent cluster kernels needed every CU - members waited for residency until the exchange time-outs: 1.3 s per step)
        self._mark("decoder head bwd")
        # ---- LSTM2 -> LSTM1 -> attention RNN loop (software-pipelined over time chunks when clusters are active)
        g2, cn2, cs2, hs2 = ctx["l2"]
        g1, cn1, cs1, hs1 = ctx["l1"]
        h1, att_out = ctx["h1"], ctx["att_out"]
        ag, acn, acs, ahs = ctx["att_saved"]
        Cn, cws1, cws2 = ctx["cluster"]
        lp1, lp2 = self.lstm_cluster_packs(Cn) if Cn else (None, None)
        Ca, aws = ctx["att_cluster"]
        NC = ctx["chunks"]
        dxg, dxg1 = self._e(1, Md, 4 * D), self._e(1, Md, 4 * D)
        dh1, datt = self._e(Md, D), self._e(Md, A + CT)
        dxga, dctx, dpq = self._e(Md, 4 * A), self._e(Md, CT), self._e(Md, U1 + U2)
        dkeys1, dkeys2 = self._e(M, U1), (self._e(M, U2) if c.dual else None)
        de1, de2 = self._e(B, Td, Ti), self._e(B, Td, Ti)
        ctx["_de"] = (de1, de2)         # (kept for diagnosis tools: tools/probes/saf_determinism3.py)
        pg_acc = None
        if "saf" in ctx:
            need = ops.attn_param_grads_acc_doubles(ctx["att_params"])      # one float64 slot per workgroup: grows with B * Ti
            if self.__dict__.get("_pg_acc") is None or self._pg_acc.numel() < need:
                self._pg_acc = ops.attn_param_grads_acc_buffer(ctx["att_params"], self.dev)
            pg_acc = self._pg_acc
        Fn = c.att_filters
        dfl = self._e(Md * Ti, Fn)
        attn_kw = dict(WrecT=self.shadow["att.Wrec.T"], WqT=self.shadow["att.Wq.T"], dout=datt,
                       dalign1=ctx.get("dalign1"), dalign2=ctx.get("dalign2"), dxg=dxga, dctx=dctx, dpq=dpq,
                       de1=de1, de2=de2, dfl=dfl)
        dzag = None
        if c.transition_agent:
            dzag = self._e(Md, 1)
            attn_kw["dz"] = dzag

        def lstm2_dw(direct=False):       # direct: on the current stream instead of the weight-gradient streams
            run = (lambda f: f()) if direct else self._wgrad
            run(lambda: (ops.linear_dw(h1, dxg[0], G["dec.lstm2.W"][:D], db=G["dec.lstm2.b"])))
            run(lambda: (ops.shifted_dw(hs2[0], Td, -1, dxg[0], G["dec.lstm2.W"][D:])))

        def lstm1_dw(direct=False):
            run = (lambda f: f()) if direct else self._wgrad
            run(lambda: (ops.linear_dw(att_out, dxg1[0], G["dec.lstm1.W"][:A + CT], db=G["dec.lstm1.b"])))
            run(lambda: (ops.shifted_dw(hs1[0], Td, -1, dxg1[0], G["dec.lstm1.W"][A + CT:])))

        if NC > 1:
            main = ops.current_stream()
            s1, s2 = self._streams()
            bounds = self._chunk_bounds(Td, NC)
            bst1, bst2 = self._e(B, 2, D), self._e(B, 2, D)
            ast = ops.attn_cluster_state(ctx["att_params"], Ca, self.dev)
            # ONE attention-backward launch over all chunks (see forward()): the kernel waits (bounded, in-kernel) for the
            # word `ready` that the LSTM1 stream bumps after each chunk's incoming gradients exist, and counts its finished
            # chunks in `done`, on which the deferred parameter gradients wait
            single = self.single_launch_attention and len(bounds) <= 16 and self.overlap_wgrad and \
                ops.streams_run_concurrently(main, s1) and ops.streams_run_concurrently(main, s2)
            if single:
                cnt = self._ctr[self._ctr_par][16:48]
                ready, done = cnt[0:1], cnt[16:32]
                # The kernel signals `done` at the pipeline-chunk boundaries AND at extra points inside the chunks it
                # processes last (<= 40 steps apart): the deferred gradients of a piece can start as soon as the piece
                # is done, so less of that work is left when the loop ends.  `ready` counts the kernel's pieces: after
                # pipeline chunk k the producer writes the number of pieces up to and including chunk k.
                pieces, pieces_upto = self._backward_pieces(bounds, Td)
            ctx["single_launch_bwd"] = bool(single)
            # (with a split head, see above: the recurrent streams start behind the suffix rows' event; chunks that reach below
            # the split row wait for the prefix launch as well - it runs on the main stream in front of the attention kernel)
            hs = self._head_split
            ev0 = hs[1] if hs else torch.cuda.Event()
            if not hs:
                ev0.record(main)
            ev_low = hs[2] if hs else None
            # the low tiles of the split head (rows < low_rows): released onto the weight-gradient stream two chunks before the
            # pipeline reaches them - by then the LSTM stream is a few hundred microseconds ahead of the attention kernel
            low_rows, low_fn, ev_low2 = (hs[3], hs[4], None) if hs else (0, None, None)
            order = list(reversed(bounds))
            k_rel = max(0, next(i for i, (b0, _) in enumerate(order) if b0 < low_rows) - max(1, self.head_split_release)) if low_fn else -1
            if single:
                self._side_mark("attention backward launch: main stream reaches it")
                with self._t("attn_rnn_bwd"):
                    ops.attn_cluster_bwd(ctx["att_params"], Ca, self._pack_cache[Ca][1], aws, 0, Td, None, ready=ready,
                                         done=done, bounds=[b0 for (b0, _) in pieces], **attn_kw)
            first = True
            pg_done = False
            pg_chunks = []
            for k, (t0, t1) in enumerate(reversed(bounds)):
                with ops.on_stream(s2):
                    if first:
                        s2.wait_event(ev0)
                        if single and self.residency and self.residency.get("attention_first"):
                            # (r6) the attention kernel is resident as a whole before the first LSTM launch may spread over the CUs
                            # it needs: csrc/attn_cluster.hip counts its resident workgroups in the word behind `ready`
                            ops.stream_wait_value(cnt[1:2], B * Ca, s2)
                        self._side_mark("LSTM 2 backward, first chunk: released (its stream)")
                    if ev_low is not None and t0 < hs[0]:
                        s2.wait_event(ev_low); ev_low = None
                    if ev_low2 is not None and t0 < low_rows:
                        s2.wait_event(ev_low2); ev_low2 = None
                    with self._t("lstm2_bwd"):
                        ops.lstm_cluster_bwd(ddec, lp2[1], B, Td, D, Cn, training, c.zc, c.zh, seed,
                                             S_L2_C, S_L2_H, g2, cn2, cs2, dxg, cws2, t0, t1, bst2)
                    ops.linear_dx_rows(dxg[0], self.W("dec.lstm2.W").rows(0, D), dh1, B, Td, t0, t1)
                    if first:
                        self._side_mark("LSTM 2 backward, first chunk: done")
                    if s1 is not s2:
                        e2 = torch.cuda.Event(); e2.record(s2)
                with ops.on_stream(s1):
                    if first and s1 is not s2:
                        s1.wait_event(ev0)
                    if s1 is not s2:
                        s1.wait_event(e2)
                    with self._t("lstm1_bwd"):
                        ops.lstm_cluster_bwd(dh1, lp1[1], B, Td, D, Cn, training, c.zc, c.zh, seed,
                                             S_L1_C, S_L1_H, g1, cn1, cs1, dxg1, cws1, t0, t1, bst1)
                    ops.linear_dx_rows(dxg1[0], self.W("dec.lstm1.W").rows(0, A + CT), datt, B, Td, t0, t1)
                    if single:
                        ops.stream_write_value(ready, pieces_upto[k], s1)     # chunk k of d att_out exists
                        if first:
                            self._side_mark("LSTM 1 backward, first chunk: done (the attention kernel's first `ready`)")
                    else:
                        e1 = torch.cuda.Event(); e1.record(s1)
                if not single:
                    main.wait_event(e1)
                    with self._t("attn_rnn_bwd"):
                        ops.attn_cluster_bwd(ctx["att_params"], Ca, self._pack_cache[Ca][1], aws, t0, t1, ast, **attn_kw)
                    if self.overlap_wgrad:
                        evc = torch.cuda.Event(); evc.record(main)
                        pg_chunks.append((t0, t1, evc))
                if k == k_rel:
                    if self.head_split_low == 2:        # IN the LSTM stream, between two chunks (no other cluster launch of that stream is
                        with ops.on_stream(s1):         # resident then: the fused backward gets the half of the chip the LSTM layers use)
                            s1.wait_event(hs[2])
                            low_fn[0]()
                            if s2 is not s1:
                                ev_low2 = torch.cuda.Event(); ev_low2.record(s1)
                            self._wgrad(low_fn[1])
                    else:                               # on the weight-gradient stream, beside the loop
                        if self._wg_stream is None:
                            self._wg_stream = self._device_streams(self.dev)[2]
                        wg = self._wg_stream
                        er = torch.cuda.Event(); er.record(s1)
                        wg.wait_event(er); wg.wait_event(hs[2])
                        with ops.on_stream(wg):
                            low_fn[0](); low_fn[1]()
                            ev_low2 = torch.cuda.Event(); ev_low2.record(wg)
                        if self._wg_used is None:
                            self._wg_used = []
                        if wg not in self._wg_used:
                            self._wg_used.append(wg)
                first = False
            if single:
                pg_chunks = [(p0, p1, r) for r, (p0, p1) in enumerate(pieces)]
                with ops.on_stream(s1):
                    e1 = torch.cuda.Event(); e1.record(s1)
            # The deferred (non-recurrent) attention gradients run chunk by chunk as the attention backward completes
            # them, on the WEIGHT-GRADIENT stream: it is idle for most of the loop, whereas the LSTM2 stream is busy
            # with its chunks for the first 2 ms and then cannot catch up before the loop ends (3.1 ms of work).  The
            # weight gradients of the two LSTMs take the LSTM streams instead, which idle once their chains are done
            # (ROCm multiplexes streams onto 4 hardware queues: a fifth stream would serialise with one of these).
            # The LDS pad keeps the workgroups on CUs the recurrent kernels do not occupy.
            pgs = s2
            if self.overlap_wgrad:
                if self._wg_stream is None:
                    self._wg_stream = self._device_streams(self.dev)[2]
                pgs = self._wg_stream
            with ops.on_stream(s2):
                lstm2_dw(direct=pgs is not s2)
                e2 = torch.cuda.Event(); e2.record(s2)
            with ops.on_stream(pgs):
                if pgs is not s2:
                    pgs.wait_event(ev0)
                # the small chunks at the start of the backward loop are merged into one launch: a 16-step launch of
                # this kernel costs 17 us per step, a 100-step launch 6 us (fixed per-workgroup set-up), and the stream
                # must keep pace with the attention loop (10.5 us per step) or its backlog lands behind the loop
                merged = self._merge_leading(pg_chunks, Td)
                for i, (t0, t1, evc) in enumerate(merged):
                    if single:
                        ops.stream_wait_value(done[evc:evc + 1], B * Ca, pgs)     # evc = piece index here
                    else:
                        pgs.wait_event(evc)
                    # the last piece starts when the recurrent kernels are gone: no LDS pad, all CUs
                    pad = 0 if (i == len(merged) - 1 and len(merged) > 1) else self.pg_lds_pad
                    with self._t("attn_param_grads"):
                        if pg_acc is not None:      # saved factors: float64 accumulators for the heavily cancelling parameter sums
                            ops.attn_param_grads_acc(ctx["att_params"], de1, de2, dkeys1, dkeys2, pg_acc, t0, t1,
                                                     accumulate=pg_done, lds_pad=pad)
                        else:
                            ops.attn_param_grads(ctx["att_params"], de1, de2, dkeys1, dkeys2, G["dec.att1.v"],
                                                 G["dec.att1.b"], G["dec.att1.U"], G.get("dec.att2.v"), t0, t1,
                                                 accumulate=pg_done, lds_pad=pad)
                    pg_done = True
                if pg_done:      # d keys are complete HERE: the memory gradients wait for this event, not for the finish launch below
                    evp = torch.cuda.Event(enable_timing=self.marks is not None); evp.record(pgs)
                    self._pg_mark = evp
                if pg_done and pg_acc is not None:
                    # (parameter gradients only: ordered before the optimiser by the join of the weight-gradient streams)
                    ops.attn_param_grads_finish(ctx["att_params"], pg_acc, G["dec.att1.v"], G["dec.att1.b"], G["dec.att1.U"],
                                                G.get("dec.att2.v"))
                    if self._wg_used is None:
                        self._wg_used = []
                    if pgs not in self._wg_used:
                        self._wg_used.append(pgs)
            with ops.on_stream(s1):
                lstm1_dw(direct=pgs is not s2)
                e1 = torch.cuda.Event(); e1.record(s1)
            self._pg_ev = evp if pg_done else None   # waited for right before the first use of d keys
            self._join = (e1, e2)
            if self.overlap_wgrad and self._wg_stream is not None:
                self._wg_rr = [self._wg_stream, s1, s2]     # the pipeline streams are idle from here on
        else:
            with self._t("lstm2_bwd"):
                if Cn:
                    ops.lstm_cluster_bwd(ddec, lp2[1], B, Td, D, Cn, training, c.zc, c.zh, seed,
                                         S_L2_C, S_L2_H, g2, cn2, cs2, dxg, cws2)
                else:
                    ops.lstm_bwd(ddec, self.shadow["l2.Wh.T"], None, 1, B, Td, D, training, c.zc, c.zh, seed,
                                 (S_L2_C,), (S_L2_H,), g2, cn2, cs2, dxg)
            lstm2_dw()
            ops.linear_dx(dxg[0], self.W("dec.lstm2.W").rows(0, D), dh1)
            with self._t("lstm1_bwd"):
                if Cn:
                    ops.lstm_cluster_bwd(dh1, lp1[1], B, Td, D, Cn, training, c.zc, c.zh, seed,
                                         S_L1_C, S_L1_H, g1, cn1, cs1, dxg1, cws1)
                else:
                    ops.lstm_bwd(dh1, self.shadow["l1.Wh.T"], None, 1, B, Td, D, training, c.zc, c.zh, seed,
                                 (S_L1_C,), (S_L1_H,), g1, cn1, cs1, dxg1)
            lstm1_dw()
            ops.linear_dx(dxg1[0], self.W("dec.lstm1.W").rows(0, A + CT), datt)
            with self._t("attn_rnn_bwd"):
                if Ca:
                    ops.attn_cluster_bwd(ctx["att_params"], Ca, self._pack_cache[Ca][1], aws, **attn_kw)
                else:
                    ops.attn_rnn_bwd(ctx["att_params"], **attn_kw)
            self._join = None
            pg_done = False
        self._mark("decoder loop bwd")
        # gradients that are plain sums over steps: recomputed massively parallel, outside the serial loop
        if not pg_done:
            with self._t("attn_param_grads"):
                if pg_acc is not None:
                    ops.attn_param_grads_acc(ctx["att_params"], de1, de2, dkeys1, dkeys2, pg_acc)
                    ops.attn_param_grads_finish(ctx["att_params"], pg_acc, G["dec.att1.v"], G["dec.att1.b"], G["dec.att1.U"],
                                                G.get("dec.att2.v"))
                else:
                    ops.attn_param_grads(ctx["att_params"], de1, de2, dkeys1, dkeys2, G["dec.att1.v"], G["dec.att1.b"],
                                         G["dec.att1.U"], G.get("dec.att2.v"))
        # memories: dvalues = align^T dctx + dkeys Wm^T ; dWm = values^T dkeys.  The two sources are independent until
        # d lstm_out is summed: with two sources the first one's chain runs on a pipeline stream (idle since the recurrent
        # loop drained) beside the second source's chain + the encoder self-attention backward on this stream.  Issued BEFORE the
        # decoder weight gradients below so that those queue behind it on the side streams, not in front of it.
        pg_ev, self._pg_ev = self._pg_ev, None      # the deferred attention gradients (d keys) of the last chunk, on their own stream
        dlstm_out = self._e(M, V1)

        def source1():
            dv1 = self._e(M, V1)
            ops.gemm(Ti, V1, Td, ctx["al1"], Ti, dctx, CT, 1, dv1, V1, a_mode=1, batch=(B, 1),
                     sA=(Td * Ti, 0), sB=(Td * CT, 0), sC=(Ti * V1, 0))
            if pg_ev is not None:
                ops.current_stream().wait_event(pg_ev)
            ops.linear_dx(dkeys1, self.W("dec.att1.Wm"), dv1, accumulate=True)
            self._wgrad(lambda: (ops.linear_dw(ctx["values1"], dkeys1, G["dec.att1.Wm"])))
            ops.seq_mask(dv1, slen, dlstm_out, B, Ti, V1)
        src1_done = None
        if c.dual and self.overlap_wgrad and self._wg_rr is not None and len(self._wg_rr) > 1:
            side = self._wg_rr[-1]       # the second pipeline stream: its last work is the deferred attention gradients
            ev = torch.cuda.Event(); ev.record(ops.current_stream())
            side.wait_event(ev)
            with ops.on_stream(side):
                source1()
                src1_done = torch.cuda.Event(); src1_done.record(side)
        else:
            source1()
        if c.dual:
            dv2 = self._e(M, V2)
            ops.gemm(Ti, V2, Td, ctx["al2"], Ti, dctx[:, V1:], CT, 1, dv2, V2, a_mode=1, batch=(B, 1),
                     sA=(Td * Ti, 0), sB=(Td * CT, 0), sC=(Ti * V2, 0))
            if pg_ev is not None:
                ops.current_stream().wait_event(pg_ev)
            ops.linear_dx(dkeys2, self.W("dec.att2.Wm"), dv2, accumulate=True)
            self._wgrad(lambda: (ops.linear_dw(ctx["values2"], dkeys2, G["dec.att2.Wm"])), defer=True)
            dsa_out = self._e(M, V2)
            ops.seq_mask(dv2, slen, dsa_out, B, Ti, V2)
        # location filter: dF[j,0,k] = sum a_{t-1}[t'+j-pl] * dfl[t',k]  (a 1-channel conv weight gradient), dbF
        aprev = torch.empty(B, Td * Ti, dtype=torch.float32, device=self.dev)
        self._keep.append(aprev)

        def loc_filter_dw():
            # location-conv input of step t: the softmax alignments of step t-1, or their running sum (cumulative_weights)
            conv_in = ctx["acum"] if c.cumulative_weights else ctx["a1"]
            if ops.loc_filter_dw(conv_in, dfl, G["dec.att1.F"], G["dec.att1.bF"], B, Td, Ti, c.att_kernel, c.att_filters):
                return
            # other filter shapes: a 1-channel conv weight gradient through the GEMM (shifted copy of the alignments)
            aprev[:, :Ti].zero_()
            if Td > 1:
                ops.axpby(conv_in.view(B, Td * Ti)[:, :(Td - 1) * Ti], aprev[:, Ti:], 1.0, 0.0)
            ops.conv1d_dw(aprev.view(Md * Ti, 1), Ti, dfl, G["dec.att1.F"], splitk=max(1, min(1024, (Md * Ti) // 4096)))
            ops.colsum(dfl, G["dec.att1.bF"])
        self._wgrad(loc_filter_dw, defer=True)
        pn = c.dec_prenet[-1]
        dpre = ctx["dpre"]
        Wa, Ga = P["dec.att_lstm.W"], G["dec.att_lstm.W"]
        self._wgrad(lambda: (ops.linear_dw(dpre[-1], dxga, Ga[:pn], db=G["dec.att_lstm.b"])), defer=True)
        self._wgrad(lambda: (ops.shifted_dw(att_out[:, A:], Td, -1, dxga, Ga[pn:pn + CT])), defer=True)
        self._wgrad(lambda: (ops.shifted_dw(ahs, Td, -1, dxga, Ga[pn + CT:])), defer=True)
        self._wgrad(lambda: (ops.linear_dw(att_out[:, :A], dpq, G["dec.att.Wq"])), defer=True)
        if c.transition_agent:      # d agent weights: sums over steps of d z [ctx1 | processed query 1] (and of d z for the bias)
            pqs = ctx["pq"]
            self._wgrad(lambda: (ops.linear_dw(att_out[:, A:A + V1], dzag, G["dec.att1.Wa"][:V1], db=G["dec.att1.ba"])), defer=True)
            self._wgrad(lambda: (ops.linear_dw(pqs[:, :U1], dzag, G["dec.att1.Wa"][V1:])), defer=True)
        # ---- decoder pre-net: only parameter gradients come out of it (the teacher-forced inputs need none), so the
        #      whole chain runs on the weight-gradient stream, off the critical path to the encoder backward
        def dec_prenet_bwd():
            dx = self._e(Md, pn)
            ops.linear_dx(dxga, self.W("dec.att_lstm.W").rows(0, pn), dx)
            xin = [ctx["dec_in"]] + dpre
            spk = ctx.get("spk")
            for n in reversed(range(len(c.dec_prenet))):
                dp = self._e(Md, c.dec_prenet[n])
                _, sc = ops.rate_thresh(self.dec_prenet_rate(training, plain=not (n == 0 and spk is not None)))
                ops.act_bwd(dx, dpre[n], dp, ACT_RELU, sc)
                if n == 0 and spk is not None:
                    self._wgrad(lambda: (ops.linear_dw(spk["d0"], dp, G["dec.prenet0.W2"], db=G["dec.prenet0.b2"])))
                    dd0 = self._e(Md, c.dec_prenet[0])
                    ops.linear_dx(dp, self.W("dec.prenet0.W2"), dd0)
                    ds = self._e(B, c.dec_prenet[0])
                    ops.segment_colsum(dd0, ds, B, Td, c.dec_prenet[0])
                    dsp = self._e(B, c.dec_prenet[0])
                    ops.act_bwd(ds, spk["sproj"], dsp, ACT_SOFTSIGN)
                    self._wgrad(lambda: (ops.linear_dw(spk["semb"], dsp, G["dec.prenet0.Ws"], db=G["dec.prenet0.bs"])))
                    dsemb = self._e(B, c.speaker_dim)
                    ops.linear_dx(dsp, self.W("dec.prenet0.Ws"), dsemb)
                    ops.embedding_bwd(ctx["batch"]["speaker_id"], dsemb, G["speaker_embedding"], offset=c.speaker_offset)
                    dr0 = self._e(Md, c.dec_prenet[0])
                    ops.act_bwd(dd0, spk["r0"], dr0, ACT_RELU)
                    self._wgrad(lambda: (ops.linear_dw(ctx["dec_in"], dr0, G["dec.prenet0.W"], db=G["dec.prenet0.b"])))
                    continue
                self._wgrad(lambda: (ops.linear_dw(xin[n], dp, G[f"dec.prenet{n}.W"], db=G[f"dec.prenet{n}.b"])))
                if n > 0:
                    dx = self._e(Md, c.dec_prenet[n - 1])
                    ops.linear_dx(dp, self.W(f"dec.prenet{n}.W"), dx)
        self._wgrad(dec_prenet_bwd, defer=True)
        self._wgrad_flush()        # the decoder's weight gradients: one event for the whole list
        if on_decoder_grads_ready is not None:
            # every decoder-parameter gradient has been ISSUED: order the callback (DP bucket all-reduce) after all
            # of them on the weight-gradient stream, without blocking the main stream's encoder backward
            self._fire_behind_wgrads(on_decoder_grads_ready)

        self._mark("memory gradients")
        # ---- encoder
        H = c.cbhg_out_units // 2
        if c.dual:
            dsa_in = dsa_out
            for h in reversed(range(c.sa_num_hop)):
                dsa_in = self._mha_bwd(dsa_in, sa_prefix("enc.sa", h), B, Ti, c.sa_units, c.sa_heads, False,
                                       Drop(rate(c.sa_drop), S_ENC_SA + HOP_STREAM * h, seed), ctx[sa_prefix("enc_mha", h)])
            lstm_out = ctx["lstm_out"]
            self._wgrad(lambda: (ops.linear_dw(lstm_out, dsa_in, G["enc.sa_proj.W"], db=G["enc.sa_proj.b"])), defer=True)
            if src1_done is not None:
                ops.current_stream().wait_event(src1_done)
            ops.linear_dx(dsa_in, self.W("enc.sa_proj.W"), dlstm_out, accumulate=True)
        self._mark("encoder self-attention bwd")
        eg, ecn, ecs, ehs = ctx["enc_lstm"]
        dxge = self._e(2, M, 4 * H)
        with self._t("enc_lstm_bwd"):
            ops.lstm_bwd(dlstm_out, self.shadow["enc.Wh.T"], slen, 2, B, Ti, H, training, c.zc, c.zh, seed,
                         (S_ENC_FW_C, S_ENC_BW_C), (S_ENC_FW_H, S_ENC_BW_H), eg, ecn, ecs, dxge)
        self._mark("encoder LSTM bwd")
        hws, zs = ctx["hws"], ctx["zs"]
        dhw = self._e(M, H)
        for d, nme in enumerate(("fw", "bw")):
            Gw = G[f"enc.lstm_{nme}.W"]
            self._wgrad(lambda d=d, nme=nme, Gw=Gw: (ops.linear_dw(hws[-1], dxge[d], Gw[:H], db=G[f"enc.lstm_{nme}.b"])), defer=True)
            self._wgrad(lambda d=d, Gw=Gw: (ops.shifted_dw(ehs[d], Ti, -1 if d == 0 else 1, dxge[d], Gw[H:])), defer=True)
            ops.linear_dx(dxge[d], self.W(f"enc.lstm_{nme}.W").rows(0, H), dhw, accumulate=(d == 1))
        self._wgrad_flush()        # self-attention block and BiLSTM weight gradients: one event, beside the highway chain
        hwW = [self.W(f"enc.highway{n}.W") for n in range(c.num_highway)]
        if self.fused_highway and ops.highway_stack_ok(hwW, H):
            dzs, dxd = [self._e(M, 2 * H) for _ in hwW], self._e(M, H)
            ops.highway_stack_bwd(dhw, hws[0], hwW, zs, hws[1:], dzs, dxd)
            for n in range(c.num_highway):
                self._wgrad(lambda n=n: (ops.linear_dw(hws[n], dzs[n], G[f"enc.highway{n}.W"], db=G[f"enc.highway{n}.b"])), defer=True)
            dhw = dxd
        else:
            for n in reversed(range(c.num_highway)):
                dz, dxd = self._e(M, 2 * H), self._e(M, H)
                ops.highway_bwd(dhw, zs[n], hws[n], dz, dxd)
                self._wgrad(lambda n=n, dz=dz: (ops.linear_dw(hws[n], dz, G[f"enc.highway{n}.W"], db=G[f"enc.highway{n}.b"])), defer=True)
                ops.linear_dx(dz, hwW[n], dxd, accumulate=True)
                dhw = dxd
        self._mark("highway bwd")
        # dhw = gradient wrt (proj2_bn + prenet_out)
        bn_st = ctx["bn_st"]
        p1 = ctx["pre"][-1]
        CC, K = c.conv_channels, c.max_filter_width
        nb = CC * K

        def bn_b(dyv, xp, name, act):
            mean, rstd, ws = bn_st[name]
            dxp = self._e(*xp.shape)
            if ws is None and ops.bn_bwd_fused(dyv, xp, P[f"enc.{name}.gamma"], P[f"enc.{name}.beta"], mean, rstd, dxp,
                                               G[f"enc.{name}.gamma"], G[f"enc.{name}.beta"],
                                               self._bn_fused_state(name, xp.shape[0], xp.shape[1]), act):
                return dxp
            if ws is None:
                ws = ops.bn_ws(xp.shape[0], xp.shape[1], self.dev)
            ops.bn_bwd(dyv, xp, P[f"enc.{name}.gamma"], P[f"enc.{name}.beta"], mean, rstd, dxp,
                       G[f"enc.{name}.gamma"], G[f"enc.{name}.beta"], ws, act)
            return dxp
        dpr2_pre = bn_b(dhw, ctx["pr2_pre"], "proj2", ACT_NONE)
        self._wgrad(lambda: (ops.conv1d_dw(ctx["pr1"], Ti, dpr2_pre, G["enc.proj2.W"])), defer=True)
        dpr1 = self._e(M, c.proj1)
        ops.conv1d_dx(dpr2_pre, Ti, self.W("enc.proj2.W"), dpr1)
        dpr1_pre = bn_b(dpr1, ctx["pr1_pre"], "proj1", ACT_RELU)
        self._wgrad(lambda: (ops.conv1d_dw(ctx["mp"], Ti, dpr1_pre, G["enc.proj1.W"])), defer=True)
        self._wgrad_flush()        # highway, proj2 and proj1 weight gradients: one event
        if on_upper_encoder_grads_ready is not None:
            self._fire_behind_wgrads(on_upper_encoder_grads_ready)
        dmp = self._e(M, nb)
        ops.conv1d_dx(dpr1_pre, Ti, self.W("enc.proj1.W"), dmp)
        if ctx["bank"] is None:         # fused forward (bn + relu + max-pool): the backward recomputes the activated bank
            mean, rstd, ws = bn_st["bank"]
            dbank_pre = self._e(M, nb)
            ops.maxpool_bn_bwd(dmp, ctx["bank_pre"], P["enc.bank.gamma"], P["enc.bank.beta"], mean, rstd, dbank_pre,
                               G["enc.bank.gamma"], G["enc.bank.beta"], ws, self._e(M, nb), B, Ti, ACT_RELU)
        else:
            dbank = self._e(M, nb)
            ops.maxpool_bwd(dmp, ctx["bank"], dbank, B, Ti, nb)
            dbank_pre = bn_b(dbank, ctx["bank_pre"], "bank", ACT_RELU)
        self._mark("projections + pool bwd")
        dp1 = dhw   # residual branch gradient; conv-bank gradients accumulate on top
        fused = self._bank_contiguous()
        if fused:
            # weight gradients of all widths in one launch (their tensors are contiguous in the flat gradient buffer)
            o1 = self.layout["enc.bank1.W"][0]
            nbw = CC * p1.shape[1] * (K * (K + 1) // 2)
            gbank = self.grad[o1:o1 + nbw]
            self._wgrad(lambda: ops.conv_bank_dw(p1, Ti, dbank_pre, gbank, K))      # (the largest one: launched right away)
            ops.conv_bank_dx(dbank_pre, Ti, self.W("enc.bank1.W"), c.max_filter_width, dp1)
        else:
            for k in range(1, c.max_filter_width + 1):
                sl = dbank_pre[:, (k - 1) * CC:k * CC]
                self._wgrad(lambda sl=sl, k=k: (ops.conv1d_dw(p1, Ti, sl, G[f"enc.bank{k}.W"])), defer=True)
                ops.conv1d_dx(sl, Ti, self.W(f"enc.bank{k}.W"), dp1, accumulate=True)
        self._mark("conv bank bwd")
        # ---- encoder pre-net + embedding
        xin = [ctx["emb"]] + ctx["pre"]
        dx = dp1
        for n in reversed(range(len(c.enc_prenet))):
            dp = self._e(M, c.enc_prenet[n])
            _, sc = ops.rate_thresh(rate(c.enc_prenet_drop))
            ops.act_bwd(dx, ctx["pre"][n], dp, ACT_RELU, sc)
            self._wgrad(lambda n=n, dp=dp: (ops.linear_dw(xin[n], dp, G[f"enc.prenet{n}.W"], db=G[f"enc.prenet{n}.b"])))
            dx = self._e(M, xin[n].shape[1])
            ops.linear_dx(dp, self.W(f"enc.prenet{n}.W"), dx)
        ops.embedding_bwd(ctx["batch"]["source"], dx, G["embedding"])
        self._mark("pre-net + embedding bwd")
        self._wgrad_join()
        for e in (self._join or ()):
            ops.current_stream().wait_event(e)
        self._join = None
        self._keep = None
        self._mark("weight-gradient join")

    # ------------------------------------------------------------------ optimiser
    def zero_grad(self):
        self.grad.zero_()

    def optimizer_step(self, grad_scale=1.0):
        """clip_by_global_norm(1.0) + TF-Adam + Noam schedule, fused over the flat buffer; then refresh the
        bf16 shadows of the recurrent weights (models/models.py:485-498, :594-598)."""
        h = self.hyper
        ops.sumsq(self.grad, self.opt_state)
        # the sticky error words of the cluster workspaces this engine has used: a hand-off timeout in ANY step since the last
        # host check leaves garbage gradients - the update is then skipped on the device (check_clusters raises later)
        errs = [ops.cluster_err_word(ws) for ws in (self._ws_last or {}).values()]
        ops.adam_step(self.flat, self.grad, self.m, self.v, self.opt_state, self.step_dev, self.seed, h["lr0"],
                      h["decay"], h["step_factor"], h["b1"], h["b2"], h["eps"], h["clip"], grad_scale, err_words=errs)
        self._refresh_shadows_async()
        self.global_step += 1

    global_step = 0     # host mirror of the optimiser step counter (the device copy drives the schedule)

    def learning_rate(self):
        """the rate the NEXT update will use: Noam-style decay of the reference (models/models.py:594-598)"""
        h = self.hyper
        if not h["decay"]:
            return float(h["lr0"])
        warm = 4000.0
        step = self.global_step * h["step_factor"] + 1
        return float(h["lr0"] * warm ** 0.5 * min(step * warm ** -1.5, step ** -0.5))

    _shadow_ev = None
    _pg_ev = None
    _loss_ev = None
    _ctr = None          # [2, 48] int32: chunk counters of the single-launch attention kernels (fwd 16 | bwd 32), two-block ring
    _ctr_par = 0
    _ctr_ev = None

    def _refresh_shadows_async(self):
        """bf16 shadows of the recurrent weights on the weight-gradient stream (idle here): the conversions overlap the
        head of the next step's encoder instead of sitting between two steps; consumers call _wait_shadows()."""
        if not self.overlap_wgrad:
            self.refresh_shadows()
            return
        if self._wg_stream is None:
            self._wg_stream = self._device_streams(self.dev)[2]
        ev = torch.cuda.Event(); ev.record(ops.current_stream())
        self._wg_stream.wait_event(ev)
        with ops.on_stream(self._wg_stream):
            def first():
                self._shadow_ev = torch.cuda.Event(); self._shadow_ev.record(self._wg_stream)
            self.refresh_shadows(after_gemm_shadows=first)
            self._shadow_ev2 = torch.cuda.Event(); self._shadow_ev2.record(self._wg_stream)

    _shadow_ev2 = None

    def _wait_shadows(self):
        """the GEMM operand shadows (needed by the first GEMM of a step)"""
        if self._shadow_ev is not None:
            ops.current_stream().wait_event(self._shadow_ev)
            self._shadow_ev = None

    def _wait_recurrent_shadows(self):
        """the recurrent kernels' operand layouts and cluster packs (first needed by the encoder LSTM)"""
        self._wait_shadows()
        if self._shadow_ev2 is not None:
            ops.current_stream().wait_event(self._shadow_ev2)
            self._shadow_ev2 = None

    def train_step(self, batch, allreduce=None):
        """One teacher-forced optimisation step.  `allreduce(lo, hi)` (optional) sums self.grad[lo:hi] across
        data-parallel ranks; it is called per bucket as soon as the bucket's gradients are final."""
        self.zero_grad()
        ctx = self.forward(batch, training=True)
        if allreduce is not None:
            # gradient buckets in the order the hand-written backward finishes them (reference train.py:68,74: MirroredStrategy's
            # one all-reduce per step): decoder (10.4 MB, under the whole encoder backward), upper encoder (enc.proj1.W .. enc.sa,
            # ~5 MB, under the conv-bank backward), conv bank + pre-net + embedding (9.6 MB: the exposed tail).  dp_buckets = 2
            # keeps the encoder in one bucket (the round-3 plan: no collective queued in front of the conv-bank weight gradients).
            mid = self.enc_mid if self.dp_buckets >= 3 else None
            self.backward(ctx, on_decoder_grads_ready=lambda: allreduce(self.enc_end, self.nparam),
                          on_upper_encoder_grads_ready=(lambda: allreduce(mid, self.enc_end)) if mid else None)
            # a rank whose cluster kernels timed out must not skip its update ALONE (the others would apply the summed garbage
            # and the replicas diverge for good): it poisons the last bucket, the sum is non-finite on every rank, and
            # satt_adam_step skips on all of them (every error word of the step is final here: the backward has been issued)
            errs = [ops.cluster_err_word(ws) for ws in (self._ws_last or {}).values()]
            if errs:
                ops.poison_on_error(self.grad, errs)
            allreduce(0, mid if mid else self.enc_end)
        else:
            self.backward(ctx)
        return ctx

    def to_device_batch(self, batch, lease=None):
        """host batch dict -> device tensors.  Arrays in page-locked memory (datasets.ljspeech.PinnedRing) are uploaded
        asynchronously on the current stream; ordinary arrays with a blocking copy.  lease = (ring, slot) - or the batch's own
        `.pinned` attribute (datasets.ljspeech.PinnedBatch) - : an event is recorded behind the copies and the ring waits for it before it refills the slot (the
        host enqueues several steps ahead of the GPU: without the fence the prefetch thread could overwrite buffers a queued
        copy has not read yet; until r5 only the blocking copies of the pageable length arrays kept that from happening)."""
        out = {}
        lease = lease or getattr(batch, "pinned", None)
        for k, v in batch.items():
            t = torch.as_tensor(v)
            if t.dtype in (torch.float64, torch.float32):
                t = t.to(torch.float32)
            pinned = self.dev.type == "cuda" and t.device.type == "cpu" and t.is_pinned()
            out[k] = t.to(self.dev, non_blocking=pinned).contiguous()
        if lease is not None and self.dev.type == "cuda":
            lease[0].uploaded(lease[1])
        return out
