"""Model selection and Estimator-style drivers with the reference's names (reference models/models.py:1180-1381):
`encoder_factory`, `decoder_factory`, `tacotron_model_factory` -> an object with `train / evaluate / predict`.

The reference builds TF graphs inside `model_fn`; this build has ONE hand-written engine (engine.Engine) for the
paths `DualSourceSelfAttentionTacotronModel` = `SelfAttentionCBHGEncoder` + `DualSourceTransformerDecoder` and the
baseline `ExtendedTacotronV1Model` = `ZoneoutEncoderV1` (zoneout CBHG) + `ExtendedDecoder` (both decoder v2),
so the factories VALIDATE a configuration against what the kernels implement and hand back small descriptors:
an unknown string raises the reference's `ValueError` (models/models.py:1254,1359,1380), a known-but-unbuilt one raises
`UnsupportedConfiguration` (a ValueError) - nothing is silently replaced by the dual-source model.
`predict` yields the reference's prediction dict (models/models.py:566-588): id, key, mel, [mel_postnet],
ground_truth_mel, alignment, alignment2, alignment5.., source, text, alignments laid out [T_memory, T_query]."""
import glob
import logging
import os
from collections import namedtuple

import numpy as np
import torch

from ..modules.attentions import UnsupportedConfiguration
from ..modules.module import DecoderModule, EncoderModule
from .attention_factories import attention_factory, dual_source_attention_factory

ENCODERS = ("SelfAttentionCBHGEncoderWithAccentType", "SelfAttentionCBHGEncoder", "EncoderV1WithAccentType",
            "ZoneoutEncoderV1", "EncoderV2")
DECODERS = ("ExtendedDecoder", "TransformerDecoder", "DualSourceDecoder", "DualSourceTransformerDecoder", "MgcLf0Decoder",
            "MgcLf0DualSourceDecoder", "DualSourceMgcLf0TransformerDecoder")
MODELS = ("MgcLf0TacotronModel", "DualSourceSelfAttentionMgcLf0TacotronModel", "DualSourceSelfAttentionTacotronModel",
          "ExtendedTacotronV1Model")


class NanLossDuringTrainingError(FloatingPointError):
    """a training step produced a non-finite loss / gradient (the name tf.estimator raises under)"""


class EncoderSpec(namedtuple("EncoderSpec", ["name", "is_training", "cbhg_out_units", "conv_channels", "max_filter_width",
                                             "projection1_out_channels", "projection2_out_channels", "num_highway",
                                             "self_attention_out_units", "self_attention_num_heads", "prenet_out_units",
                                             "drop_rate", "zoneout_factor_cell", "zoneout_factor_output",
                                             "self_attention_drop_rate", "self_attention_num_hop"]), EncoderModule):
    """what encoder_factory returns: the hyper-parameters (a namedtuple) AND the reference's call contract
    `encoder(inputs, input_lengths=...) -> (lstm_output, self_attention_output, alignments)` (modules/module.py EncoderModule)"""


class DecoderSpec(namedtuple("DecoderSpec", ["name", "prenet_out_units", "drop_rate", "attention_rnn_out_units",
                                             "decoder_version", "decoder_out_units", "num_mels", "outputs_per_step", "max_iters",
                                             "n_feed_frame", "zoneout_factor_cell", "zoneout_factor_output",
                                             "self_attention_out_units", "self_attention_num_heads", "self_attention_drop_rate",
                                             "self_attention_num_hop"]), DecoderModule):
    """what decoder_factory returns: hyper-parameters + `decoder(source, attention1_fn=..., ...) -> (mel, stop_token, state)`"""



def encoder_factory(params, is_training):
    """reference models/models.py:1180-1255"""
    if params.encoder not in ENCODERS or (params.encoder == "EncoderV1WithAccentType" and not params.use_accent_type) \
            or (params.encoder == "ZoneoutEncoderV1" and params.use_accent_type):
        raise ValueError(f"Unknown encoder: {params.encoder}")
    if params.encoder == "ZoneoutEncoderV1":       # reference models/models.py:1232-1243, modules/module.py:293-342
        if not params.use_zoneout_at_encoder:
            raise UnsupportedConfiguration("ZoneoutEncoderV1 with use_zoneout_at_encoder=False (plain CBHG with a GRU, "
                                           "module.py:321-329) is not built; the shipped tacotron.json sets it True")
        return EncoderSpec(params.encoder, is_training, params.cbhg_out_units, params.conv_channels, params.max_filter_width,
                           params.projection1_out_channels, params.projection2_out_channels, params.num_highway,
                           0, 0, tuple(params.encoder_prenet_out_units), params.encoder_prenet_drop_rate,
                           params.zoneout_factor_cell, params.zoneout_factor_output, 0.0, 0)._with_params(params)
    if params.encoder != "SelfAttentionCBHGEncoder":
        raise UnsupportedConfiguration(f"encoder {params.encoder} is not built for MI355X (only SelfAttentionCBHGEncoder, "
                                       "modules/module.py:374-441, and ZoneoutEncoderV1, :293-342)")
    if params.self_attention_num_hop < 1:
        raise ValueError("self_attention_num_hop must be >= 1")
    return EncoderSpec(params.encoder, is_training, params.cbhg_out_units, params.conv_channels, params.max_filter_width,
                       params.projection1_out_channels, params.projection2_out_channels, params.num_highway,
                       params.self_attention_out_units, params.self_attention_num_heads,
                       tuple(params.encoder_prenet_out_units), params.encoder_prenet_drop_rate,
                       params.zoneout_factor_cell, params.zoneout_factor_output, params.self_attention_drop_rate,
                       params.self_attention_num_hop)._with_params(params)


def decoder_factory(params):
    """reference models/models.py:1258-1360"""
    if params.decoder not in DECODERS:
        raise ValueError(f"Unknown decoder: {params.decoder}")
    if params.decoder not in ("DualSourceTransformerDecoder", "ExtendedDecoder"):
        raise UnsupportedConfiguration(f"decoder {params.decoder} is not built for MI355X (only "
                                       "DualSourceTransformerDecoder, modules/module.py:1449-1559, and ExtendedDecoder, "
                                       ":530-623)")
    if params.decoder_version != "v2":
        raise UnsupportedConfiguration(f"decoder_version {params.decoder_version}: only v2 (DecoderRNNV2: two ZoneoutLSTM "
                                       "layers, modules/module.py:1527-1534) is built; v1 stacks residual GRU cells")
    if params.decoder == "ExtendedDecoder":         # reference models/models.py:1258-1270: no self-attention block
        return DecoderSpec(params.decoder, tuple(params.decoder_prenet_out_units), params.decoder_prenet_drop_rate,
                           params.attention_out_units, params.decoder_version, params.decoder_out_units, params.num_mels,
                           params.outputs_per_step, params.max_iters, params.n_feed_frame, params.zoneout_factor_cell,
                           params.zoneout_factor_output, 0, 0, 0.0, 0)._with_params(params)
    if params.decoder_self_attention_num_hop < 1:
        raise ValueError("decoder_self_attention_num_hop must be >= 1")
    return DecoderSpec(params.decoder, tuple(params.decoder_prenet_out_units), params.decoder_prenet_drop_rate,
                       params.attention_out_units, params.decoder_version, params.decoder_out_units, params.num_mels,
                       params.outputs_per_step, params.max_iters, params.n_feed_frame, params.zoneout_factor_cell,
                       params.zoneout_factor_output, params.decoder_self_attention_out_units,
                       params.decoder_self_attention_num_heads, params.decoder_self_attention_drop_rate,
                       params.decoder_self_attention_num_hop)._with_params(params)


def validate_params(params):
    """everything the reference reads on this path that the kernels do not implement fails HERE, loudly.
    Returns (encoder spec, decoder spec, attention1_fn, attention2_fn)."""
    if params.tacotron_model not in MODELS:
        raise ValueError(f"Unknown Tacotron model: {params.tacotron_model}")
    if params.tacotron_model not in ("DualSourceSelfAttentionTacotronModel", "ExtendedTacotronV1Model"):
        raise UnsupportedConfiguration(f"tacotron_model {params.tacotron_model} is not built for MI355X (only "
                                       "DualSourceSelfAttentionTacotronModel, models/models.py:229-588, and "
                                       "ExtendedTacotronV1Model, :20-226)")
    enc = encoder_factory(params, True)
    dec = decoder_factory(params)
    baseline = params.tacotron_model == "ExtendedTacotronV1Model"
    # the reference wires the single-source model_fn to any encoder / decoder pair; the pairs the kernels implement:
    want = ("ZoneoutEncoderV1", "ExtendedDecoder") if baseline else ("SelfAttentionCBHGEncoder", "DualSourceTransformerDecoder")
    if (enc.name, dec.name) != want:
        raise UnsupportedConfiguration(f"{params.tacotron_model} is built with encoder={want[0]}, decoder={want[1]} "
                                       f"(got {enc.name}, {dec.name})")
    if baseline:
        a1, a2 = attention_factory(params), None
    else:
        a1, a2 = dual_source_attention_factory(params)
    if a1.options.attention not in ("forward", "location_sensitive"):
        raise UnsupportedConfiguration(f"attention={a1.options.attention}: the first source needs a location-aware "
                                       "mechanism (forward or location_sensitive)")
    if a2 is not None and a2.options.attention != "additive":
        raise UnsupportedConfiguration(f"attention2={a2.options.attention}: only additive (BahdanauAttention) is built "
                                       "for the second source")
    for flag in ("use_accent_type", "use_external_speaker_embedding", "speaker_embedd_to_decoder",
                 "speaker_embedd_to_postnet", "channel_id_to_postnet", "use_language_embedding"):
        if getattr(params, flag):
            raise UnsupportedConfiguration(f"{flag}=True is not built for MI355X")
    if params.use_speaker_embedding and not params.speaker_embedd_to_prenet:
        raise UnsupportedConfiguration("use_speaker_embedding needs speaker_embedd_to_prenet=True (MultiSpeakerPreNet)")
    if params.spec_loss_type not in ("l1", "mse"):
        raise ValueError(f"Unknown loss type: {params.spec_loss_type}")
    return enc, dec, a1, a2


class RunConfig(namedtuple("RunConfig", ["save_summary_steps", "save_checkpoints_steps", "keep_checkpoint_max",
                                         "log_step_count_steps"])):
    """the fields of tf.estimator.RunConfig the reference sets (train.py:60-74)"""

    @classmethod
    def from_hparams(cls, hp):
        return cls(hp.save_summary_steps, hp.save_checkpoints_steps, hp.keep_checkpoint_max, hp.log_step_count_steps)


class _StepProfiler:
    """one optimisation step under torch.profiler (CPU + GPU activities; on ROCm the GPU side comes from roctracer), exported
    as a Chrome trace - the format of the reference's tf.train.ProfilerHook timelines.  For per-kernel counters use
    `rocprofv3 --kernel-trace --stats -- python train.py ...` instead (profiles/README.md)."""

    def __init__(self, path):
        self.path, self.p = path, None

    def start(self):
        from torch.profiler import ProfilerActivity, profile
        self.p = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])
        self.p.__enter__()

    def stop(self):
        torch.cuda.synchronize()
        self.p.__exit__(None, None, None)
        self.p.export_chrome_trace(self.path)
        logging.info("profile of one train step written to %s", self.path)


def _batches(input_fn):
    ds = input_fn() if callable(input_fn) else input_fn
    return getattr(ds, "dataset", ds)


def _tensor_items(batch):
    return {k: v for k, v in batch.items() if hasattr(v, "dtype") and k != "id"}


class DualSourceSelfAttentionTacotronModel:
    """Estimator-shaped wrapper of the training / evaluation / synthesis drivers around engine.Engine
    (reference models/models.py:229-588 + tf.estimator.Estimator's train / evaluate / predict)."""
    MODEL_NAME = "DualSourceSelfAttentionTacotronModel"

    def __init__(self, params, model_dir, config=None, warm_start_from=None, device=None, dp=None, rng_seed=None):
        from ..engine import Engine
        from ..params import ModelConfig
        self.params = params
        self.model_dir = model_dir
        self.config = config or RunConfig.from_hparams(params)
        if params.tacotron_model != self.MODEL_NAME:
            raise ValueError(f"{type(self).__name__} built from hparams of {params.tacotron_model}")
        self.encoder_spec, self.decoder_spec, self.attention1_fn, self.attention2_fn = validate_params(params)
        self.dp = dp
        rank = dp.rank if dp is not None else 0
        self.engine = Engine(ModelConfig.from_hparams(params), device or "cuda", lr0=params.initial_learning_rate,
                             decay=params.decay_learning_rate, step_factor=params.learning_rate_step_factor,
                             b1=params.adam_beta1, b2=params.adam_beta2, eps=params.adam_eps,
                             loss_type=params.spec_loss_type, rng_seed=rank if rng_seed is None else rng_seed)
        self.encoder_spec.bind(self.engine); self.decoder_spec.bind(self.engine)      # the callable modules share the model's parameters
        self.global_step = 0
        if model_dir:
            os.makedirs(model_dir, exist_ok=True)
            self.restore_latest()
        self.warm_started = []
        # Initialisation only (tf.estimator semantics): explicit WarmStartSettings always; the hparams switch only for a TRAINING
        # model (model_dir given) - predict_mel.py builds the model without a model_dir and restores its own checkpoint, exactly
        # as the reference's predict never passes warm-start settings (train.py:76-78 builds them, predict_mel.py does not)
        if (warm_start_from is not None or (params.warm_start and model_dir)) and not (model_dir and self.checkpoint_paths()):
            # tf.estimator semantics (reference train.py:76-78): initialisation only - a checkpoint in model_dir wins
            from .warm_start import load_var_map, warm_start
            ckpt = getattr(warm_start_from, "ckpt_to_initialize_from", warm_start_from) or params.ckpt_to_initialize_from
            pats = getattr(warm_start_from, "vars_to_warm_start", None) or params.vars_to_warm_start
            if not ckpt:
                raise ValueError("warm_start=True needs ckpt_to_initialize_from")
            if os.path.isdir(ckpt):      # a model directory: its `checkpoint` state file names the latest checkpoint
                state = os.path.join(ckpt, "checkpoint")
                line = open(state).readline() if os.path.exists(state) else ""
                if "model_checkpoint_path" not in line:
                    raise ValueError("no `checkpoint` state file in %s" % ckpt)
                ckpt = os.path.join(ckpt, line.split(":", 1)[1].strip().strip('"'))
            vmap = load_var_map(params.warm_start_var_map) if params.warm_start_var_map else None
            self.warm_started = warm_start(self.engine, ckpt, pats, vmap)
            logging.info("warm start: %d variables from %s", len(self.warm_started), ckpt)

    # ------------------------------------------------------------------ checkpoints (tf.estimator: model_dir)
    def checkpoint_paths(self):
        ps = glob.glob(os.path.join(self.model_dir, "model-*.pt"))
        return sorted(ps, key=lambda p: int(p.rsplit("-", 1)[1][:-3]))

    def restore(self, path):
        eng = self.engine
        st = torch.load(path, map_location="cpu")
        if st["params"].numel() != eng.nparam:
            raise ValueError("checkpoint %s holds %d parameter floats, this build's flat layout has %d: the parameter layout "
                             "(params.layout: tensor order / alignment padding) or the model hparams changed since it was written"
                             % (path, st["params"].numel(), eng.nparam))
        eng.flat.copy_(st["params"])
        if "m" in st:
            eng.m.copy_(st["m"]); eng.v.copy_(st["v"])
        for k, (m, v) in st.get("bn", {}).items():
            eng.bn[k][0].copy_(m); eng.bn[k][1].copy_(v)
        self.global_step = int(st.get("step", 0))
        eng.global_step = self.global_step
        eng.step_dev.fill_(self.global_step)
        if "seed" in st:          # the dropout / zoneout counter the optimiser kernel advances: resume the mask sequence
            eng.seed.fill_(int(st["seed"]))
        eng.refresh_shadows()
        return self.global_step

    def restore_latest(self):
        ps = self.checkpoint_paths()
        return self.restore(ps[-1]) if ps else 0

    def save(self):
        eng = self.engine
        path = os.path.join(self.model_dir, "model-%d.pt" % self.global_step)
        torch.save({"step": self.global_step, "params": eng.flat.cpu(), "m": eng.m.cpu(), "v": eng.v.cpu(),
                    "bn": {k: (m.cpu(), v.cpu()) for k, (m, v) in eng.bn.items()}, "seed": int(eng.seed.cpu()[0])}, path)
        keep = int(self.config.keep_checkpoint_max or 0)
        if keep > 0:
            for old in self.checkpoint_paths()[:-keep]:
                os.remove(old)
        return path

    # ------------------------------------------------------------------ train
    def train(self, input_fn, steps=None, max_steps=None, writer=None, on_checkpoint=None):
        """tf.estimator.Estimator.train: `steps` more optimiser steps, or until global step `max_steps`.
        input_fn() -> iterable of padded batch dicts (datasets.*.DatasetSource...group_by_batch())."""
        eng, dp, cfg = self.engine, self.dp, self.config
        world = dp.world if dp is not None else 1
        rank = dp.rank if dp is not None else 0
        if dp is not None:
            dp.bind(eng.grad)
            dp.broadcast_params(eng.flat)
            eng.refresh_shadows()
        stop_at = max_steps if max_steps is not None else (self.global_step + steps if steps is not None else None)
        ctx = None
        hp = self.params
        saver = None
        if rank == 0 and self.model_dir:         # reference models/models.py:499-508 (MetricsSaver training hook)
            from ..utils.metrics_saver import MetricsSaver
            saver = MetricsSaver(self.model_dir, hp.alignment_save_steps, "train", hp.save_training_time_metrics,
                                 hp.keep_eval_results_max_epoch)
        prof = None
        last_nan_check = self.global_step
        eng.opt_state[-2:].zero_()               # the device's sticky skip counters (csrc/elementwise.hip adam_prepare_k)
        for batch in _batches(input_fn):
            if stop_at is not None and self.global_step >= stop_at:
                break
            b = eng.to_device_batch(_tensor_items(batch), lease=getattr(batch, "pinned", None))
            step = self.global_step + 1
            if rank == 0 and self.model_dir and hp.record_profile and step % max(1, hp.profile_steps) == 0:
                # reference models/models.py:510-513: tf.train.ProfilerHook(save_steps=profile_steps, output_dir=model_dir) writes
                # timeline-<step>.json (Chrome trace format); here the kernel timeline of this one step through torch.profiler
                prof = _StepProfiler(os.path.join(self.model_dir, "timeline-%d.json" % step))
                prof.start()
            ctx = eng.train_step(b, allreduce=dp.allreduce if world > 1 else None)
            if dp is not None:
                dp.wait()
            log_now = step % max(1, cfg.log_step_count_steps) == 0
            # rank-independent predicate: the MAX-over-ranks below is a collective, every rank must reach it on the same steps
            # (only the checkpoint WRITE is rank 0's)
            ckpt_step = step % max(1, cfg.save_checkpoints_steps) == 0
            ckpt_now = rank == 0 and ckpt_step
            if log_now or ckpt_step:
                # a timed-out cluster hand-off leaves garbage gradients: never let it into Adam.  The device-side guard has skipped
                # the updates of the affected EARLIER steps on every rank (satt_poison_on_error makes the skip global); THIS step's
                # update is skipped below, before the error words are cleared.  First occurrence: fall back to the event-ordered
                # chunk schedule and keep training; a second one - on the fallback schedule - is raised.  Under data parallelism
                # the decision is taken by all ranks together (MAX over ranks of the local flag): replicas must stay identical.
                err = None
                try:
                    eng.check_clusters(ctx)
                except RuntimeError as e:
                    err = e
                bad = err is not None
                if dp is not None and world > 1:
                    bad = dp.max_over_ranks(1.0 if bad else 0.0) > 0.0
                if bad:
                    if not eng.recover_from_handoff_timeout():
                        raise err if err is not None else RuntimeError("cluster hand-off timeout on another rank")
                    print("[train] WARNING step %d: %s - this update and those since the last check were skipped; continuing on "
                          "the chunk-by-chunk attention schedule" % (step, err or "hand-off timeout on another rank"), flush=True)
                    if prof is not None:
                        prof.stop(); prof = None
                    # (the steps skipped for the time-out - on a peer: through its poisoned bucket - are accounted for: the sticky
                    #  counters start again, on every rank alike)
                    eng.opt_state[-2:].zero_()
                    continue                      # no optimizer_step for this batch: its gradients are the garbage
            eng.optimizer_step(grad_scale=1.0 / world)
            self.global_step = step
            if log_now or ckpt_step:
                # the device-side guard (adam_prepare_k) counts the updates it skipped for a non-finite gradient alone (sticky: the
                # count covers EVERY step since the last look, not only this one), and the cluster check above was clean.
                # tf.estimator stops at the first such step (NanLossDuringTrainingError, NanTensorHook of the reference's
                # Estimator); here the run stops at the next log / checkpoint step - before anything is saved - and says how many
                # updates were dropped.  The all-reduced gradient - and so this count - is the same on every rank: all raise together
                nan_skips = int(float(eng.opt_state[-1]))
                if nan_skips:
                    raise NanLossDuringTrainingError("step %d: non-finite loss / gradient in %d update(s) since step %d - skipped on the "
                                                     "device, never applied" % (step, nan_skips, last_nan_check))
                last_nan_check = step
            if prof is not None:
                prof.stop(); prof = None
            if saver is not None and saver.due(step):
                o = eng.outputs(ctx)
                als = [o["alignment1"].float().cpu().numpy()]
                if eng.cfg.dual:
                    als.append(o["alignment2"].float().cpu().numpy())
                    ea = o["enc_alignment"].float().cpu().numpy()
                    als += [ea[:, h] for h in range(ea.shape[1])]
                Bn = b["source"].shape[0]
                saver.save(step, batch.get("id", np.arange(Bn)), batch.get("key", [str(i) for i in range(Bn)]),
                           batch.get("text", [""] * Bn), np.asarray(batch["source"]), np.asarray(batch["source_length"]), als,
                           o["mel"].float().cpu().numpy(), np.asarray(batch["mel"]), np.asarray(batch["target_length"]),
                           r=eng.cfg.r)
            if log_now and rank == 0:
                ls = [float(x) for x in eng.losses.cpu()]
                logging.info("step %d loss %.5f mel_loss %.5f done_loss %.5f", step, ls[2], ls[0], ls[1])
                if writer is not None:
                    sc = {"mel_loss": ls[0], "done_loss": ls[1], "loss": ls[2], "learning_rate": eng.learning_rate()}
                    if eng.cfg.l2_weight > 0:       # reference models/models.py:247-248
                        sc["l2_regularization_loss"] = float(eng.reg_loss.cpu()[0])
                    writer.add_scalars(step, sc)
                    writer.flush()
            if ckpt_now:
                path = self.save()
                if on_checkpoint is not None:
                    on_checkpoint(step, path)
            if dp is not None and world > 1 and step % max(1, cfg.save_checkpoints_steps) == 0:
                # rank 0 has just saved and evaluated: the other ranks wait HERE, at an explicit barrier with the process group's
                # raised timeout, not inside the next step's gradient all-reduce under the collective watchdog
                dp.barrier()
        if ctx is not None:
            eng.check_clusters(ctx)
        return self

    # ------------------------------------------------------------------ evaluate (models/models.py:517-564)
    def evaluate(self, input_fn, steps=None):
        from ..inference import evaluate
        from ..utils.summary import EVAL_SCALARS
        acc, n = {}, 0
        hp = self.params
        saver = None
        if self.model_dir:        # EVAL-mode MetricsSaver (reference models/models.py:552-562): results of the free run
            from ..utils.metrics_saver import MetricsSaver
            saver = MetricsSaver(os.path.join(self.model_dir, "eval"), hp.alignment_save_steps, "eval",
                                 keep_eval_results_max_epoch=hp.keep_eval_results_max_epoch)
        for batch in _batches(input_fn):
            if steps is not None and n >= steps:
                break
            ev = evaluate(self.engine, _tensor_items(batch))
            for k in EVAL_SCALARS:
                acc[k] = acc.get(k, 0.0) + ev[k]
            if saver is not None and n == 0:          # one batch per evaluation run, like the hook's first eval step
                Bn = len(batch["source"])
                als = [ev["alignment1"].float().cpu().numpy()]
                if self.engine.cfg.dual and ev.get("alignment2") is not None:
                    als.append(ev["alignment2"].float().cpu().numpy())
                saver.save(self.global_step, batch.get("id", np.arange(Bn)), batch.get("key", [str(i) for i in range(Bn)]),
                           batch.get("text", [""] * Bn), np.asarray(batch["source"]), np.asarray(batch["source_length"]), als,
                           ev["mel"].float().cpu().numpy(), np.asarray(batch["mel"]), np.asarray(batch["target_length"]),
                           r=self.engine.cfg.r)
            n += 1
        out = {k: v / n for k, v in acc.items()} if n else {}
        out["global_step"] = self.global_step
        return out

    # ------------------------------------------------------------------ predict (models/models.py:566-588)
    def predict(self, input_fn):
        """yields ONE prediction dict per utterance, keys and layouts of the reference's EstimatorSpec predictions"""
        from ..inference import infer, postnet_infer
        eng, hp = self.engine, self.params
        r = hp.outputs_per_step
        for batch in _batches(input_fn):
            b = eng.to_device_batch(_tensor_items(batch))
            spk = b.get("speaker_id")
            B = b["source"].shape[0]
            if hp.use_forced_alignment_mode:
                # pass 1 (models/models.py:387-390): validation decode fed with the GROUND TRUTH mel, exactly Td steps;
                # pass 2 (:411-428): free feeding, both mechanisms return pass 1's alignments (TeacherForcing*Attention)
                if "mel" not in b:
                    raise ValueError("use_forced_alignment_mode needs the target mel (--target-data-root)")
                first = infer(eng, b["source"], b["source_length"], teacher=b["mel"], speaker_id=spk)
                out = infer(eng, b["source"], b["source_length"], max_steps=first["steps"], speaker_id=spk,
                            min_steps=1 << 30, teacher_alignments=(first["alignment1"], first["alignment2"] if eng.cfg.dual else None))
            else:
                out = infer(eng, b["source"], b["source_length"], max_steps=hp.max_iters, speaker_id=spk)
            mel_post = postnet_infer(eng, out["mel"]) if eng.cfg.use_postnet_v2 else None
            enc_al = out.get("enc_alignment")
            for i in range(B):
                p = {"id": int(batch["id"][i]) if "id" in batch else i, "key": batch["key"][i] if "key" in batch else str(i),
                     "mel": out["mel"][i].float().cpu().numpy()}
                if mel_post is not None:
                    p["mel_postnet"] = mel_post[i].float().cpu().numpy()
                if "mel" in batch:
                    p["ground_truth_mel"] = np.asarray(batch["mel"][i])
                p["alignment"] = out["alignment1"][i].cpu().numpy().T          # [T_memory, T_query]
                if eng.cfg.dual:                 # the single-source model_fn has no second history (models.py:196-212)
                    p["alignment2"] = out["alignment2"][i].cpu().numpy().T
                if enc_al is not None:                                          # encoder self-attention heads
                    for h in range(enc_al.shape[1]):
                        p["alignment%d" % (5 + h)] = enc_al[i, h].cpu().numpy().T
                p["source"] = np.asarray(batch["source"][i])
                p["text"] = batch["text"][i] if "text" in batch else ""
                yield p


class ExtendedTacotronV1Model(DualSourceSelfAttentionTacotronModel):
    """the baseline Tacotron (reference models/models.py:20-226; examples/*/tacotron.json): ZoneoutEncoderV1 +
    ExtendedDecoder v2 - the same engine with one attention source and no self-attention blocks; train / evaluate /
    predict as above (its prediction dict has no alignment2.. keys, :196-212)."""
    MODEL_NAME = "ExtendedTacotronV1Model"


def tacotron_model_factory(hparams, model_dir, run_config=None, warm_start_from=None, **kw):
    """reference models/models.py:1363-1381"""
    if hparams.tacotron_model not in MODELS:
        raise ValueError(f"Unknown Tacotron model: {hparams.tacotron_model}")
    if hparams.tacotron_model == "ExtendedTacotronV1Model":
        return ExtendedTacotronV1Model(hparams, model_dir, config=run_config, warm_start_from=warm_start_from, **kw)
    if hparams.tacotron_model != "DualSourceSelfAttentionTacotronModel":
        raise UnsupportedConfiguration(f"tacotron_model {hparams.tacotron_model} is not built for MI355X (only "
                                       "DualSourceSelfAttentionTacotronModel and ExtendedTacotronV1Model)")
    return DualSourceSelfAttentionTacotronModel(hparams, model_dir, config=run_config, warm_start_from=warm_start_from, **kw)
