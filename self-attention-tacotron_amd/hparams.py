"""Configuration surface of the reference kept verbatim for drop-in use (reference hparams.py:10-226, consumed by
train.py:111-116): every key, name (including the historical misspelling `suffle_buffer_size`), type and default;
`parse_json(text)` then `parse("a=b,c=d")` overrides.  The reference uses tf.contrib.training.HParams; this is a
small dependency-free equivalent (TensorFlow is not part of this build)."""
import ast
import copy as _copy
import json

_AUDIO = dict(num_mels=80, num_mgcs=60, num_freq=2049, sample_rate=48000, frame_length_ms=50.0, frame_shift_ms=12.5,
              ref_level_db=20, average_mel_level_db=[0.0], stddev_mel_level_db=[0.0], min_mel_level_db=[0.0],
              silence_mel_level_db=-3.0, mgc_dim=60, mgc_alpha=0.77, mgc_gamma=0.0, mgc_fft_len=4096, num_lf0s=256,
              f0_max=529.0, f0_min=66.0, lf0_loss_factor=0.5)
_DATA = dict(dataset="vctk.dataset.DatasetSource", num_symbols=256, source="phoneme",
             source_file_extension="source.tfrecord", target_file_extension="target.tfrecord")
_MODEL = dict(tacotron_model="ExtendedTacotronV1Model", outputs_per_step=2, n_feed_frame=2, embedding_dim=256,
              use_accent_type=False, accent_type_embedding_dim=32, num_accent_type=129, accent_type_offset=0x3100,
              accent_type_unknown=0x3180, accent_type_prenet_out_units=(32, 16),
              encoder_prenet_out_units_if_accent=(224, 112), encoder="ZoneoutEncoderV1",
              encoder_prenet_drop_rate=0.5, cbhg_out_units=256, conv_channels=128, max_filter_width=16,
              projection1_out_channels=128, projection2_out_channels=128, num_highway=4,
              encoder_prenet_out_units=(256, 128), encoder_v2_num_conv_layers=3, encoder_v2_kernel_size=5,
              encoder_v2_out_units=512, encoder_v2_drop_rate=0.5, self_attention_out_units=32,
              self_attention_num_heads=2, self_attention_num_hop=1, self_attention_encoder_out_units=32,
              self_attention_drop_rate=0.05, self_attention_transformer_num_conv_layers=1,
              self_attention_transformer_kernel_size=5, decoder="ExtendedDecoder", attention="additive",
              forced_alignment_attention="teacher_forcing_forward", attention2="additive",
              forced_alignment_attention2="teacher_forcing_additive", attention1_out_units=224,
              attention2_out_units=32, decoder_prenet_drop_rate=0.5, apply_dropout_on_inference=False,
              decoder_prenet_out_units=(256, 128), attention_out_units=256, decoder_out_units=256,
              attention_kernel=31, attention_filters=32, cumulative_weights=False,
              use_forward_attention_transition_agent=False, decoder_self_attention_out_units=256,
              decoder_self_attention_num_heads=2, decoder_self_attention_num_hop=1,
              decoder_self_attention_drop_rate=0.05)
_SPEAKER = dict(use_speaker_embedding=False, use_external_speaker_embedding=False,
                speaker_embedding_projection_out_dim=-1, embedding_file="", num_speakers=1, speaker_embedding_dim=16,
                speaker_embedding_offset=0, speaker_for_synthesis=-1, speaker_embedd_to_prenet=True,
                speaker_embedd_to_decoder=False, speaker_embedd_to_postnet=False, channel_id_to_postnet=False,
                channel_id_file="", channel_id_dim=8, use_language_embedding=False,
                language_embedding_projection_out_dim=-1, language_embedding_file="", language_embedding_dim=16,
                language_embedd_to_input=False, language_embedd_to_decoder=False)
_POSTNET = dict(post_net_cbhg_out_units=256, post_net_conv_channels=128, post_net_max_filter_width=8,
                post_net_projection1_out_channels=256, post_net_projection2_out_channels=80, post_net_num_highway=4,
                use_postnet_v2=False, num_postnet_v2_layers=5, postnet_v2_kernel_size=5, postnet_v2_out_channels=512,
                postnet_v2_drop_rate=0.5, spec_loss_type="l1")
_TRAIN = dict(batch_size=32, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8, initial_learning_rate=0.002,
              decay_learning_rate=True, learning_rate_step_factor=1, use_l2_regularization=False,
              l2_regularization_weight=1e-7, save_summary_steps=100, save_checkpoints_steps=500,
              keep_checkpoint_max=200, keep_checkpoint_every_n_hours=1, log_step_count_steps=1,
              alignment_save_steps=10000, save_training_time_metrics=False, approx_min_target_length=100,
              suffle_buffer_size=64, batch_bucket_width=50, batch_num_buckets=50,
              interleave_cycle_length_cpu_factor=1.0, interleave_cycle_length_min=4, interleave_cycle_length_max=16,
              interleave_buffer_output_elements=200, interleave_prefetch_input_elements=200, prefetch_buffer_size=4,
              use_cache=False, cache_file_name="", logfile="log.txt", record_profile=False, profile_steps=50,
              warm_start=False, ckpt_to_initialize_from="", vars_to_warm_start=[".*"],
              # extension (not a reference hparam): JSON file mapping TensorFlow variable names onto this build's parameters,
              # required by warm_start (models/warm_start.py; the reference relies on the graph's own variable names)
              warm_start_var_map="")
_EVAL = dict(max_iters=500, num_evaluation_steps=64, keep_eval_results_max_epoch=10, eval_start_delay_secs=120,
             eval_throttle_secs=600, use_forced_alignment_mode=False, predicted_mel_extension="mfbsp",
             use_zoneout_at_encoder=False, decoder_version="v1", zoneout_factor_cell=0.1, zoneout_factor_output=0.1,
             trim_top_db=30, trim_frame_length=1024, trim_hop_length=256, num_silent_frames=4)


class HParams:
    def __init__(self, **kw):
        object.__setattr__(self, "_v", {})
        for k, v in kw.items():
            self._v[k] = v

    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, "_v")[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if k not in self._v:
            raise ValueError("Unknown hyperparameter: %s" % k)
        self._v[k] = v

    def values(self):
        return dict(self._v)

    def copy(self):
        return HParams(**_copy.deepcopy(self._v))

    @staticmethod
    def _coerce(default, value, key):
        if isinstance(value, str) and not isinstance(default, str):
            try:
                value = ast.literal_eval(value)
            except (ValueError, SyntaxError):
                raise ValueError("Could not parse value %r for hyperparameter %s" % (value, key))
        if isinstance(default, bool):
            if not isinstance(value, (bool, int)):
                raise ValueError("%s expects a bool" % key)
            return bool(value)
        if isinstance(default, float) and isinstance(value, int):
            return float(value)
        if isinstance(default, int) and not isinstance(default, bool) and isinstance(value, float) and value.is_integer():
            return int(value)
        if isinstance(default, (list, tuple)) and isinstance(value, (list, tuple)):
            return list(value)
        return value

    def set(self, key, value):
        if key not in self._v:
            raise ValueError("Unknown hyperparameter: %s" % key)
        self._v[key] = self._coerce(self._v[key], value, key)

    def parse_json(self, text):
        """JSON object (string or dict) overriding known keys (reference train.py:111-113); keys starting
        with '_' are comments."""
        d = json.loads(text) if isinstance(text, str) else dict(text)
        for k, v in d.items():
            if k.startswith("_"):
                continue
            self.set(k, v)
        return self

    def parse(self, text):
        """comma-separated name=value list; bracketed lists may contain commas (reference train.py:116)."""
        if not text:
            return self
        parts, depth, cur = [], 0, ""
        for ch in text:
            if ch in "[(":
                depth += 1
            elif ch in "])":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur); cur = ""
            else:
                cur += ch
        if cur:
            parts.append(cur)
        for p in parts:
            if "=" not in p:
                raise ValueError("Could not parse hparam %r" % p)
            k, v = p.split("=", 1)
            v = v.strip()
            if v in ("True", "true"):
                v = "True"
            elif v in ("False", "false"):
                v = "False"
            self.set(k.strip(), v)
        return self


hparams = HParams(**{**_AUDIO, **_DATA, **_MODEL, **_SPEAKER, **_POSTNET, **_TRAIN, **_EVAL})


def hparams_debug_string(hp=None):
    values = (hp or hparams).values()
    return "Hyperparameters:\n" + "\n".join("  %s: %s" % (n, values[n]) for n in sorted(values))
