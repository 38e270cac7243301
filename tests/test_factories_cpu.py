"""CPU tests of the reference-named selection surface: tacotron_model_factory / encoder_factory /
decoder_factory (reference models/models.py:1180-1381), attention_mechanism_factory (modules/attentions.py:25-62),
dual_source_attention_factory (models/attention_factories.py:22-37).  Unknown strings raise the reference's ValueError;
strings the reference knows but this build has no kernels for raise UnsupportedConfiguration (a ValueError) - nothing is
silently replaced by the dual-source self-attention model."""
import json
import os

import pytest

import satt_amd  # noqa: F401
from satt_amd.hparams import hparams as default_hparams
from satt_amd.models import attention_factories
from satt_amd.models.models import (DECODERS, ENCODERS, MODELS, decoder_factory, encoder_factory, tacotron_model_factory,
                                    validate_params)
from satt_amd.modules.attentions import AttentionOptions, UnsupportedConfiguration, attention_mechanism_factory
from satt_amd.params import ModelConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lj(name="self-attention-tacotron.json", corpus="ljspeech"):
    hp = default_hparams.copy()
    d = json.load(open(os.path.join(ROOT, "examples", corpus, name)))
    d.pop("_comment", None)
    hp.parse_json(json.dumps(d))
    return hp


def test_shipped_config_resolves():
    hp = lj()
    enc, dec, a1, a2 = validate_params(hp)
    assert enc.name == "SelfAttentionCBHGEncoder" and dec.name == "DualSourceTransformerDecoder" and dec.decoder_version == "v2"
    m1 = a1("memory1", "lengths")
    m2 = a2("memory2", "lengths")
    assert (m1.kind, m1.num_units, m1.attention_kernel, m1.attention_filters, m1.cumulative_weights) == ("forward", 224, 10, 5, False)
    assert (m2.kind, m2.num_units) == ("additive", 32) and m2.memory == "memory2"
    c = ModelConfig.from_hparams(hp)
    assert c.attention == "forward" and c.cumulative_weights is False and c.att1_units == 224
    hp.parse("attention=location_sensitive,cumulative_weights=True")
    c = ModelConfig.from_hparams(hp)
    assert c.attention == "location_sensitive" and c.cumulative_weights is True


def test_model_strings():
    hp = lj()
    hp.tacotron_model = "NoSuchModel"
    with pytest.raises(ValueError, match="Unknown Tacotron model: NoSuchModel"):
        tacotron_model_factory(hp, None)
    for name in MODELS:
        if name in ("DualSourceSelfAttentionTacotronModel", "ExtendedTacotronV1Model"):
            continue
        hp.tacotron_model = name
        with pytest.raises(UnsupportedConfiguration):
            tacotron_model_factory(hp, None)
        with pytest.raises(UnsupportedConfiguration):
            ModelConfig.from_hparams(hp)
    # the reference's DEFAULT hparams select the baseline Tacotron with additive attention, a GRU CBHG and decoder v1:
    # not built, and it must not silently train something else
    with pytest.raises(UnsupportedConfiguration):
        ModelConfig.from_hparams(default_hparams.copy())
    # a model string paired with the other model's encoder / decoder is refused as well
    hp = lj(); hp.tacotron_model = "ExtendedTacotronV1Model"
    with pytest.raises(UnsupportedConfiguration):
        ModelConfig.from_hparams(hp)


def test_baseline_tacotron_config_resolves():
    """examples/ljspeech/tacotron.json (reference models/models.py:20-226): ZoneoutEncoderV1 + ExtendedDecoder v2 =
    one attention source with attention_out_units units, no self-attention parameters anywhere"""
    from satt_amd.params import layout
    hp = lj("tacotron.json")
    enc, dec, a1, a2 = validate_params(hp)
    assert (enc.name, dec.name, a2) == ("ZoneoutEncoderV1", "ExtendedDecoder", None)
    assert a1.options.num_units == hp.attention_out_units == 256 and a1.options.attention == "forward"
    c = ModelConfig.from_hparams(hp)
    assert (c.dual, c.sa_units, c.att2_units, c.dec_sa_units, c.att1_units, c.ctx_dim, c.out_in) == (False, 0, 0, 0, 256, 256, 256)
    names = set(layout(c)[0])
    assert not any(n.startswith(("enc.sa", "dec.sa", "dec.att2")) for n in names)
    assert layout(c)[0]["dec.att_lstm.W"][1] == (128 + 256 + 256, 1024) and layout(c)[0]["dec.out.W"][1] == (256, 161)
    hp.use_zoneout_at_encoder = False            # plain CBHG = GRU: not built
    with pytest.raises(UnsupportedConfiguration):
        validate_params(hp)


def test_encoder_and_decoder_strings():
    hp = lj()
    hp.encoder = "Nope"
    with pytest.raises(ValueError, match="Unknown encoder: Nope"):
        encoder_factory(hp, True)
    hp.decoder = "Nope"
    with pytest.raises(ValueError, match="Unknown decoder: Nope"):
        decoder_factory(hp)
    hp = lj()
    for name in ENCODERS:
        hp.encoder = name
        if name == "SelfAttentionCBHGEncoder":
            assert encoder_factory(hp, False).is_training is False
        elif name == "ZoneoutEncoderV1":
            assert encoder_factory(hp, True).self_attention_out_units == 0
        elif name == "EncoderV1WithAccentType":         # only valid together with use_accent_type (models/models.py:1221)
            with pytest.raises(ValueError, match="Unknown encoder"):
                encoder_factory(hp, True)
        else:
            with pytest.raises(UnsupportedConfiguration):
                encoder_factory(hp, True)
    for name in DECODERS:
        hp.decoder = name
        if name == "DualSourceTransformerDecoder":
            assert decoder_factory(hp).attention_rnn_out_units == 256
        elif name == "ExtendedDecoder":
            assert decoder_factory(hp).self_attention_out_units == 0
        else:
            with pytest.raises(UnsupportedConfiguration):
                decoder_factory(hp)
    hp = lj()
    hp.decoder_version = "v1"
    with pytest.raises(UnsupportedConfiguration):
        decoder_factory(hp)


def opts(**kw):
    base = dict(attention="forward", num_units=8, attention_kernel=3, attention_filters=2, smoothing=False,
                cumulative_weights=False, use_transition_agent=False)
    base.update(kw)
    return AttentionOptions(**base)


def test_attention_strings():
    with pytest.raises(ValueError, match="Unknown attention mechanism: dot"):
        attention_mechanism_factory(opts(attention="dot"))
    assert attention_mechanism_factory(opts(use_transition_agent=True))("mem", "len").use_transition_agent is True
    # the agent belongs to ForwardAttention only (modules/forward_attention.py:80-86): ignored by the other mechanisms
    assert attention_mechanism_factory(opts(attention="location_sensitive", use_transition_agent=True))("m", "l").use_transition_agent is False
    for name, kind in (("forward", "forward"), ("location_sensitive", "location_sensitive"), ("additive", "additive")):
        m = attention_mechanism_factory(opts(attention=name, cumulative_weights=True))("mem", "len")
        assert m.kind == kind and m.num_units == 8 and m.teacher_alignments is None
        assert m.cumulative_weights == (kind != "additive")
    fn = attention_mechanism_factory(opts(attention="teacher_forcing_forward"))
    with pytest.raises(ValueError):
        fn("mem", "len")
    assert fn("mem", "len", teacher_alignments="A").teacher_alignments == "A"
    assert attention_mechanism_factory(opts(attention="teacher_forcing_additive"))("m", "l", "A").kind == "additive"


def test_attention_factories_wiring():
    hp = lj()
    a = attention_factories.attention_factory(hp)
    assert a.options.num_units == hp.attention_out_units and a.options.attention == "forward"
    f1, f2 = attention_factories.force_alignment_dual_source_attention_factory(hp)
    assert f1.options.attention == "teacher_forcing_forward" and f2.options.attention == "teacher_forcing_additive"
    assert (f1.options.num_units, f2.options.num_units) == (224, 32)
    assert attention_factories.force_alignment_attention_factory(hp).options.num_units == 256
    hp.attention2 = "forward"
    with pytest.raises(UnsupportedConfiguration):
        validate_params(hp)
    hp = lj()
    hp.attention = "additive"
    with pytest.raises(UnsupportedConfiguration):
        validate_params(hp)
    hp = lj()
    h2 = lj(); h2.self_attention_num_hop = 2; h2.decoder_self_attention_num_hop = 3        # built (r3): stacked transformer blocks
    validate_params(h2)
    from satt_amd.params import ModelConfig, param_shapes
    c2 = ModelConfig.from_hparams(h2)
    names = [n for n, _ in param_shapes(c2)]
    assert (c2.sa_num_hop, c2.dec_sa_num_hop) == (2, 3) and "enc.sa.h1.kvq.W" in names and "dec.sa.h2.t.b" in names
    assert "enc.sa.h2.kvq.W" not in names and names.index("enc.sa.h1.t.b") < names.index("dec.prenet0.W")      # encoder bucket
    h2 = lj(); h2.apply_dropout_on_inference = True          # built (r3): decode kernels + evaluation pass keep the pre-net dropout
    validate_params(h2)
    from satt_amd.params import ModelConfig
    assert ModelConfig.from_hparams(h2).apply_dropout_on_inference is True
    for flag in ("use_accent_type", "speaker_embedd_to_decoder"):
        h2 = lj(); setattr(h2, flag, True)
        with pytest.raises(ValueError):
            validate_params(h2)
    hp.spec_loss_type = "huber"
    with pytest.raises(ValueError, match="Unknown loss type"):
        validate_params(hp)


def test_vctk_configs_resolve():
    """examples/vctk/*.json: both models with the multi-speaker decoder pre-net (152 speakers, ids from 225)"""
    for name, dual in (("self-attention-tacotron.json", True), ("tacotron.json", False)):
        c = ModelConfig.from_hparams(lj(name, "vctk"))
        assert (c.dual, c.num_speakers, c.speaker_offset) == (dual, 152, 225)
    hp = lj("tacotron.json", "vctk")
    assert hp.dataset == "vctk.dataset.DatasetSource"
    hp.speaker_embedd_to_prenet = False
    with pytest.raises(UnsupportedConfiguration):
        validate_params(hp)


def test_l2_regularization_is_the_baseline_models_only():
    """reference models/models.py:109-114 vs :278-515: only ExtendedTacotronV1Model's model_fn reads use_l2_regularization"""
    from satt_amd.params import l2_regularized
    hp = lj("tacotron.json"); hp.use_l2_regularization = True
    c = ModelConfig.from_hparams(hp)
    assert c.l2_weight == hp.l2_regularization_weight == 1e-7
    names = l2_regularized(c)
    assert "enc.bank16.W" in names and "dec.att1.v" in names and "dec.att1.Wm" in names
    assert not any(n in names for n in ("embedding", "dec.att1.b", "dec.att_lstm.W", "enc.lstm_fw.W", "dec.out.W", "enc.bank.gamma"))
    hp = lj(); hp.use_l2_regularization = True
    assert ModelConfig.from_hparams(hp).l2_weight == 0.0
