"""Callable encoder / decoder modules with the reference's call contracts (reference modules/module.py:293-342 ZoneoutEncoderV1,
:374-441 SelfAttentionCBHGEncoder, :530-623 ExtendedDecoder, :1449-1559 DualSourceTransformerDecoder).

The reference's layers are tf.layers.Layer objects that create their variables on first call; here a module is the DESCRIPTOR the
factories always returned (a namedtuple of the hyper-parameters: models/models.py EncoderSpec / DecoderSpec) plus `__call__`, thin
over the hand-written engine:

    encoder = encoder_factory(hparams, is_training)            # models/models.py:1180-1255
    lstm_out, sa_out, alignments = encoder(embedded [B, Ti, E], input_lengths=lengths)            # module.py:425
    decoder = decoder_factory(hparams)                         # models/models.py:1258-1360
    mel, stop, state = decoder((lstm_out, sa_out), attention1_fn=a1, attention2_fn=a2, speaker_embed=None, is_training=True,
                               is_validation=False, teacher_forcing=False, memory_sequence_length=lengths,
                               memory2_sequence_length=lengths, target_sequence_length=target_lengths, target=mel_targets,
                               teacher_alignments=(None, None), apply_dropout_on_inference=False)   # module.py:1493-1498

A module computes with the parameters of the Engine it is bound to: `module.bind(engine)` shares a model's engine (what
tacotron_model_factory's models do), otherwise the first call builds one from the hparams the factory was given (fresh random
parameters - the analogue of a layer creating its variables).  These calls are FORWARD passes (TF would differentiate the graph
they build; training here goes through Engine.train_step, which owns the hand-written backward)."""
import numpy as np
import torch

from .attentions import UnsupportedConfiguration


class _Bound:
    """engine binding shared by the callable modules (namedtuple subclasses: the attributes live in the instance dict)"""

    def bind(self, engine):
        self.__dict__["_engine"] = engine
        return self

    def _with_params(self, params):
        self.__dict__["_params"] = params
        return self

    @property
    def engine(self):
        eng = self.__dict__.get("_engine")
        if eng is None:
            params = self.__dict__.get("_params")
            if params is None:
                raise UnsupportedConfiguration("%s: no engine bound and no hparams to build one from" % type(self).__name__)
            from ..engine import Engine
            from ..params import ModelConfig
            eng = Engine(ModelConfig.from_hparams(params), "cuda")
            self.__dict__["_engine"] = eng
        return eng


def _dev(eng, x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(eng.dev).contiguous()


class EncoderModule(_Bound):
    """__call__(inputs, input_lengths=None) -> the reference encoders' results.  inputs: the EMBEDDED text [B, Ti, embedding_dim]
    (models/models.py:351 embeds before it calls the encoder).  SelfAttentionCBHGEncoder returns (lstm_output [B, Ti, cbhg_out_units],
    self_attention_output [B, Ti, self_attention_out_units], alignments: one [B, Ti, Ti] per head and hop - module.py:425-441);
    ZoneoutEncoderV1 returns the CBHG output alone (module.py:336-342)."""

    def __call__(self, inputs, input_lengths=None):
        eng = self.engine
        c = eng.cfg
        x = _dev(eng, inputs, torch.float32)
        if x.dim() != 3 or x.shape[2] != c.embedding_dim:
            raise ValueError("encoder inputs must be [batch, time, %d] embedded symbols (got %s)" % (c.embedding_dim, tuple(x.shape)))
        B, Ti = x.shape[:2]
        if input_lengths is None:
            input_lengths = np.full(B, Ti, np.int64)
        batch = {"embedded": x, "source_length": _dev(eng, input_lengths, torch.int64)}
        ctx = {"training": bool(self.is_training), "batch": batch}
        lstm_out, sa_out = eng._encode(batch, bool(self.is_training), ctx)
        self.__dict__["last_ctx"] = ctx
        lstm_out = lstm_out.view(B, Ti, -1)
        if not c.dual:
            return lstm_out
        aligns = [a.view(B, c.sa_heads, Ti, Ti)[:, h] for a in ctx["enc_aligns"] for h in range(c.sa_heads)]
        return lstm_out, sa_out.view(B, Ti, -1), aligns


class DecoderModule(_Bound):
    """__call__ with the keyword set of DualSourceTransformerDecoder.call / ExtendedDecoder.call -> (mel [B, Tm, num_mels],
    stop_token [B, Tm / r, 1], state).  `state` holds what the reference's final decoder state carries for model_fn
    (alignment histories: models/models.py:397-408): state["alignments"] = [alignment1 (, alignment2)], each [B, T_query, T_memory].

    Modes (module.py:1513-1526 / helpers): is_training, or is_validation with teacher_forcing -> teacher-fed pass over the padded
    target length (dropout / zoneout on only when is_training); otherwise the free-running decode of inference.infer (stop rule,
    at most max_iters steps); teacher_alignments = (a1, a2) selects the forced-alignment mode of the free run."""

    def __call__(self, source, attention1_fn=None, attention2_fn=None, speaker_embed=None, is_training=None, is_validation=None,
                 teacher_forcing=False, memory_sequence_length=None, memory2_sequence_length=None, target_sequence_length=None,
                 target=None, teacher_alignments=(None, None), apply_dropout_on_inference=None, attention_fn=None):
        eng = self.engine
        c = eng.cfg
        src = source if isinstance(source, (tuple, list)) else (source, None)
        if c.dual and (len(src) < 2 or src[1] is None):
            raise ValueError("DualSourceTransformerDecoder needs (source1, source2)")
        a1 = attention1_fn if attention1_fn is not None else attention_fn          # ExtendedDecoder.call names it attention_fn
        for fn, want in ((a1, ("forward", "location_sensitive", "teacher_forcing_forward")),
                         (attention2_fn if c.dual else None, ("additive", "teacher_forcing_additive"))):
            if fn is not None and fn.options.attention not in want:
                raise UnsupportedConfiguration("attention %s is not what this decoder's kernels were built with" % fn.options.attention)
        if a1 is not None and a1.options.attention.replace("teacher_forcing_", "") != c.attention:
            raise UnsupportedConfiguration("the bound engine runs attention=%s, the call passes %s" % (c.attention, a1.options.attention))
        if apply_dropout_on_inference is not None and bool(apply_dropout_on_inference) != bool(c.apply_dropout_on_inference):
            raise UnsupportedConfiguration("apply_dropout_on_inference is fixed when the engine is built (hparams)")
        mem1 = _dev(eng, src[0], torch.float32)
        B, Ti = mem1.shape[:2]
        mem2 = _dev(eng, src[1], torch.float32) if c.dual else None
        if memory_sequence_length is None:
            memory_sequence_length = np.full(B, Ti, np.int64)
        if c.dual and memory2_sequence_length is not None and \
                not np.array_equal(np.asarray(torch.as_tensor(memory2_sequence_length).cpu()), np.asarray(torch.as_tensor(memory_sequence_length).cpu())):
            raise UnsupportedConfiguration("memory2_sequence_length must equal memory_sequence_length (both memories come from "
                                           "one encoder: models/models.py:372-376)")
        slen = _dev(eng, memory_sequence_length, torch.int64)
        spk = {}
        if c.num_speakers > 0:
            if speaker_embed is None:
                raise ValueError("this model was built with a speaker embedding: pass speaker_embed")
            se = torch.as_tensor(speaker_embed)
            spk = {"speaker_embed": _dev(eng, se, torch.float32)} if se.is_floating_point() else {"speaker_id": _dev(eng, se, torch.int64)}
        teacher_fed = bool(is_training) or (bool(is_validation) and bool(teacher_forcing))
        mechs = self._mechanisms(a1, attention2_fn, (mem1, mem2), slen, teacher_alignments)
        if teacher_fed:
            if target is None:
                raise ValueError("teacher-fed decoding needs target")
            tgt = _dev(eng, target, torch.float32)
            Tm = tgt.shape[1]
            if Tm % c.r:
                raise ValueError("target length must be a multiple of outputs_per_step")
            tl = _dev(eng, target_sequence_length if target_sequence_length is not None else np.full(B, Tm, np.int64), torch.int64)
            Td = Tm // c.r
            ones = torch.ones(B, Tm, dtype=torch.float32, device=eng.dev)
            batch = dict(encoder_outputs=(mem1, mem2), source_length=slen, mel=tgt, target_length=tl,
                         done=torch.zeros(B, Td, dtype=torch.float32, device=eng.dev), spec_loss_mask=ones,
                         binary_loss_mask=ones[:, :Td].contiguous(), **spk)
            ctx = eng.forward(batch, training=bool(is_training))
            out = eng.outputs(ctx)
            als = [out["alignment1"]] + ([out["alignment2"]] if c.dual else [])
            mel, stop = out["mel"], out["stop"]
            self.__dict__["last_ctx"] = ctx
        else:
            from ..inference import infer
            forced = teacher_alignments is not None and teacher_alignments[0] is not None
            kw = {}
            if "speaker_embed" in spk:
                kw["speaker_embed"] = spk["speaker_embed"]
            elif "speaker_id" in spk:
                kw["speaker_id"] = spk["speaker_id"]
            steps = self.max_iters
            if forced:
                steps = min(steps, int(torch.as_tensor(teacher_alignments[0]).shape[1]))
            res = infer(eng, torch.zeros(B, Ti, dtype=torch.int64, device=eng.dev), slen, max_steps=steps,
                        teacher_alignments=tuple(teacher_alignments) if forced else None, encoder_outputs=(mem1, mem2), **kw)
            als = [res["alignment1"]] + ([res["alignment2"]] if c.dual else [])
            mel, stop = res["mel"], res["stop"]
        for m, a in zip(mechs, als):
            m.__dict__["alignments"] = a
        return mel, stop, {"alignments": als, "attention_mechanisms": mechs}

    @staticmethod
    def _mechanisms(a1, a2, mems, slen, teacher_alignments):
        out = []
        ta = teacher_alignments if teacher_alignments is not None else (None, None)
        for fn, mem, t in ((a1, mems[0], ta[0]), (a2, mems[1], ta[1] if len(ta) > 1 else None)):
            if fn is not None and mem is not None:
                teach = t if fn.options.attention.startswith("teacher_forcing_") else None
                out.append(fn(mem, slen, teach) if teach is not None else fn(mem, slen))
        return out
