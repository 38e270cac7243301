"""Attention selection surface of the reference (modules/attentions.py:14-62): `AttentionOptions` and
`attention_mechanism_factory(options)` -> `attention_fn(memory, memory_sequence_length, teacher_alignments=None)`.

The reference's closures build TF attention objects; here a mechanism is a small DESCRIPTOR that the MI355X engine
resolves into flags of the fused attention-RNN kernels (csrc/attn_cluster.hip, csrc/attn_rnn.hip).  Same strings, same
`ValueError` for an unknown name (modules/attentions.py:59), and a loud `UnsupportedConfiguration` - never a silent
substitute - for names the reference knows but this build has no kernel for."""
from collections import namedtuple


class UnsupportedConfiguration(ValueError):
    """a value the reference accepts but the MI355X kernels do not implement"""


class AttentionOptions(namedtuple("AttentionOptions", ["attention", "num_units", "attention_kernel",
                                                       "attention_filters", "smoothing", "cumulative_weights",
                                                       "use_transition_agent"])):
    pass


# kernel-level kinds (satt_attn_rnn_params.att1_kind)
KIND_FORWARD, KIND_LOCATION_SENSITIVE, KIND_ADDITIVE = "forward", "location_sensitive", "additive"
KNOWN = ("forward", "location_sensitive", "teacher_forcing_forward", "teacher_forcing_additive", "additive")


class AttentionMechanism(namedtuple("AttentionMechanism", ["kind", "num_units", "attention_kernel", "attention_filters",
                                                           "cumulative_weights", "memory", "memory_sequence_length",
                                                           "teacher_alignments", "use_transition_agent"])):
    """what `attention_fn` returns: which scoring / recursion the kernels run for this memory
    (forward: modules/forward_attention.py:88-122; location_sensitive: the same score without the alpha recursion,
    :13-26 + tacotron2 LocationSensitiveAttention; additive: tf.contrib.seq2seq.BahdanauAttention;
    teacher_forcing_*: modules/teacher_forcing_attention.py:13-78 - the given alignments are returned as they are)."""


def attention_mechanism_factory(options: AttentionOptions):
    if options.attention not in KNOWN:
        # raised when the closure is CALLED in the reference (:59); raising at construction fails earlier, never later
        raise ValueError(f"Unknown attention mechanism: {options.attention}")
    if options.attention == "location_sensitive" and options.smoothing:
        raise UnsupportedConfiguration("LocationSensitiveAttention(smoothing=True) is not built (the reference's own "
                                       "factories always pass smoothing=False, models/attention_factories.py:16,26)")

    def attention_fn(memory, memory_sequence_length, teacher_alignments=None):
        kind = options.attention
        if kind.startswith("teacher_forcing_"):
            if teacher_alignments is None:
                raise ValueError("%s needs teacher_alignments" % kind)
            kind = "forward" if kind.endswith("forward") else "additive"
        return AttentionMechanism(kind, options.num_units, options.attention_kernel, options.attention_filters,
                                  bool(options.cumulative_weights) if kind != KIND_ADDITIVE else False,
                                  memory, memory_sequence_length, teacher_alignments,
                                  # the agent exists in ForwardAttention only (modules/forward_attention.py:80-86)
                                  bool(options.use_transition_agent) and options.attention == "forward")

    attention_fn.options = options
    return attention_fn
