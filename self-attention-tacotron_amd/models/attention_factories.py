"""hparams -> attention closures, names and argument wiring of the reference (models/attention_factories.py:11-72)."""
from ..modules.attentions import AttentionOptions, attention_mechanism_factory


def _options(params, attention, num_units):
    return AttentionOptions(attention=attention, num_units=num_units, attention_kernel=params.attention_kernel,
                            attention_filters=params.attention_filters, smoothing=False,
                            cumulative_weights=params.cumulative_weights,
                            use_transition_agent=params.use_forward_attention_transition_agent)


def attention_factory(params):
    return attention_mechanism_factory(_options(params, params.attention, params.attention_out_units))


def dual_source_attention_factory(params):
    return (attention_mechanism_factory(_options(params, params.attention, params.attention1_out_units)),
            attention_mechanism_factory(_options(params, params.attention2, params.attention2_out_units)))


def force_alignment_attention_factory(params):
    return attention_mechanism_factory(_options(params, params.forced_alignment_attention, params.attention_out_units))


def force_alignment_dual_source_attention_factory(params):
    return (attention_mechanism_factory(_options(params, params.forced_alignment_attention, params.attention1_out_units)),
            attention_mechanism_factory(_options(params, params.forced_alignment_attention2, params.attention2_out_units)))
