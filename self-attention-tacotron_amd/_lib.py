"""ctypes binding of libsatt_hip.so (the C-ABI declared in include/satt_hip.h).

The product path has NO fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SATT_LIB_PATH: an alternative build of the SAME library (profile / experiment variants, tools/build_variant.sh); no fallback either way
LIB_PATH = os.environ.get("SATT_LIB_PATH") or os.path.join(_HERE, "libsatt_hip.so")

c_f32p = C.c_void_p
c_i64 = C.c_int64
c_u32 = C.c_uint32

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID, ACT_SOFTSIGN = 0, 1, 2, 3, 4
PREC_F32, PREC_BF16 = 0, 1


class GemmParams(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("nb_outer", C.c_int), ("nb_inner", C.c_int),
        ("A", C.c_void_p), ("lda", c_i64), ("strideA_o", c_i64), ("strideA_i", c_i64), ("a_mode", C.c_int),
        ("conv_T", C.c_int), ("conv_C", C.c_int), ("conv_sgn", C.c_int), ("conv_off", C.c_int),
        ("B", C.c_void_p), ("sb_tap", c_i64), ("sb_k", c_i64), ("sb_n", c_i64), ("strideB_o", c_i64),
        ("strideB_i", c_i64), ("kin", C.c_int),
        ("C", C.c_void_p), ("ldc", c_i64), ("strideC_o", c_i64), ("strideC_i", c_i64),
        ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", c_i64),
        ("act", C.c_int),
        ("alpha", C.c_float),
        ("accumulate", C.c_int),
        ("splitk", C.c_int),
        ("drop_thresh", c_u32), ("drop_scale", C.c_float), ("drop_stream", c_u32), ("seed", C.c_void_p),
        ("precision", C.c_int),
        ("bank_ng", C.c_int), ("bank_a_col", C.c_int), ("bank_c_col", C.c_int), ("bank_b_unit", c_i64),
        ("Bs", C.c_void_p), ("sbs_tap", c_i64), ("sbs_n", c_i64),
        ("colsum", C.c_void_p),
        ("ws", C.c_void_p),
    ]


class AttnRnnParams(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("Td", C.c_int), ("Ti", C.c_int),
        ("A", C.c_int), ("U1", C.c_int), ("V1", C.c_int), ("U2", C.c_int), ("V2", C.c_int),
        ("kernel", C.c_int), ("filters", C.c_int), ("training", C.c_int), ("keys_lds_bf16", C.c_int),
        ("zc", C.c_float), ("zh", C.c_float), ("zc_thresh", c_u32), ("zh_thresh", c_u32), ("seed", C.c_void_p),
        ("stream_c", c_u32), ("stream_h", c_u32),
        ("lengths", C.c_void_p), ("xg", C.c_void_p), ("Wrec", C.c_void_p), ("Wq", C.c_void_p),
        ("keys1", C.c_void_p), ("values1", C.c_void_p), ("keys2", C.c_void_p), ("values2", C.c_void_p),
        ("locF", C.c_void_p), ("locFb", C.c_void_p), ("locU", C.c_void_p), ("v1", C.c_void_p), ("b1", C.c_void_p),
        ("v2", C.c_void_p),
        ("out", C.c_void_p), ("align1", C.c_void_p), ("align2", C.c_void_p),
        ("a1", C.c_void_p), ("pq", C.c_void_p), ("fl", C.c_void_p),
        ("gates", C.c_void_p), ("cnew", C.c_void_p), ("cstate", C.c_void_p), ("hstate", C.c_void_p),
        ("teach1", C.c_void_p), ("teach2", C.c_void_p),
        ("att1_mode", C.c_int), ("cumulative", C.c_int), ("acum", C.c_void_p),
        ("agentW", C.c_void_p), ("agentb", C.c_void_p), ("ustate", C.c_void_p), ("saf", C.c_void_p),
    ]


class AttnRnnBwdParams(C.Structure):
    _fields_ = [
        ("f", AttnRnnParams),
        ("WrecT", C.c_void_p), ("WqT", C.c_void_p),
        ("dout", C.c_void_p), ("dalign1", C.c_void_p), ("dalign2", C.c_void_p),
        ("dxg", C.c_void_p), ("dctx", C.c_void_p), ("dpq", C.c_void_p),
        ("de1", C.c_void_p), ("de2", C.c_void_p), ("dfl", C.c_void_p), ("dz", C.c_void_p),
    ]


class AttnClusterParams(C.Structure):
    _fields_ = [("f", AttnRnnParams), ("C", C.c_int), ("WrecP", C.c_void_p), ("ws", C.c_void_p), ("t0", C.c_int),
                ("t1", C.c_int), ("progress", C.c_void_p), ("nbound", C.c_int), ("bound", C.c_int * 16), ("vw1", C.c_void_p)]


class AttnClusterBwdParams(C.Structure):
    _fields_ = [("b", AttnRnnBwdParams), ("C", C.c_int), ("WrecTP", C.c_void_p), ("ws", C.c_void_p), ("t0", C.c_int),
                ("t1", C.c_int), ("state", C.c_void_p), ("ready", C.c_void_p), ("done", C.c_void_p),
                ("nbound", C.c_int), ("bound", C.c_int * 16)]


class HighwayLayer(C.Structure):
    _fields_ = [("Wt", C.c_void_p), ("Wn", C.c_void_p), ("b", C.c_void_p), ("z", C.c_void_p), ("y", C.c_void_p), ("dz", C.c_void_p)]


class DecLinearParams(C.Structure):
    _fields_ = [("B", C.c_int), ("N", C.c_int), ("nseg", C.c_int),
                ("x", C.c_void_p * 3), ("x_bs", c_i64 * 3), ("x_ss", c_i64 * 3), ("k", C.c_int * 3),
                ("x_ps", c_i64 * 3), ("W", C.c_void_p), ("Wb", C.c_void_p), ("ldw", c_i64), ("bias", C.c_void_p), ("act", C.c_int),
                ("res", C.c_void_p), ("res_bs", c_i64), ("res_ss", c_i64),
                ("y", C.c_void_p), ("y_bs", c_i64), ("y_ss", c_i64), ("step", C.c_void_p),
                ("lstm_H", C.c_int), ("c_state", C.c_void_p), ("h_state", C.c_void_p), ("zc", C.c_float), ("zh", C.c_float),
                ("step_out", C.c_void_p), ("step_add", C.c_int), ("stop", C.c_void_p), ("stop_bs", c_i64), ("stop_ss", c_i64),
                ("flag", C.c_void_p), ("stop_threshold", C.c_float), ("min_steps", C.c_int),
                ("drop_thresh", C.c_uint32), ("drop_scale", C.c_float), ("drop_stream", C.c_uint32), ("drop_seed", C.c_void_p),
                ("drop_T", C.c_int)]


class DecAttentionParams(C.Structure):
    _fields_ = [("B", C.c_int), ("Td", C.c_int), ("Ti", C.c_int), ("U1", C.c_int), ("V1", C.c_int), ("U2", C.c_int),
                ("V2", C.c_int), ("kernel", C.c_int), ("filters", C.c_int), ("att1_mode", C.c_int), ("cumulative", C.c_int),
                ("A", C.c_int), ("lengths", C.c_void_p), ("hq", C.c_void_p), ("Wq", C.c_void_p), ("Wqb", C.c_void_p),
                ("pq_out", C.c_void_p), ("agentW", C.c_void_p), ("agentb", C.c_void_p),
                ("keys1", C.c_void_p), ("values1", C.c_void_p), ("keys2", C.c_void_p), ("values2", C.c_void_p),
                ("locF", C.c_void_p), ("locFb", C.c_void_p), ("locU", C.c_void_p), ("v1", C.c_void_p), ("b1", C.c_void_p),
                ("v2", C.c_void_p), ("teach1", C.c_void_p), ("teach2", C.c_void_p),
                ("a_state", C.c_void_p), ("alpha_state", C.c_void_p), ("e1", C.c_void_p), ("e2", C.c_void_p),
                ("ctx", C.c_void_p),
                ("align1", C.c_void_p), ("align2", C.c_void_p), ("step", C.c_void_p)]



class DecMegaParams(C.Structure):
    """satt_dec_mega_params (include/satt_hip.h): the persistent decode step"""
    _fields_ = ([(n, C.c_int) for n in ("B", "Td", "Ti", "A", "D", "Ds", "heads", "U1", "V1", "U2", "V2", "kernel", "filters",
                                        "att1_mode", "cumulative", "P0", "P1", "feed", "NO", "ldout")] +
                [("zc", C.c_float), ("zh", C.c_float), ("stop_threshold", C.c_float), ("min_steps", C.c_int)] +
                [(n, C.c_void_p) for n in ("Wp0", "Wp1", "Wa", "Wq", "W1", "W2", "Wkvq", "Wot", "Wout",
                                           "bp0", "bp1", "ba", "b1l", "b2l", "bkvq", "bot", "bout",
                                           "locF", "locFb", "locU", "v1", "b1", "v2", "lengths",
                                           "keys1", "values1", "keys2", "values2",
                                           "ca", "ha", "c1", "h1", "c2", "h2", "a_state", "alpha_state", "ctx", "yout", "tin",
                                           "align1", "align2", "kvq", "part", "ctab", "Wfh", "Wfl", "bfb", "step", "flag", "err")] +
                [("nsteps", C.c_int)])

# name -> (restype, argtypes); must list EVERY symbol declared in include/satt_hip.h
_P = C.c_void_p
_I = C.c_int
_F = C.c_float
SIGNATURES = {
    "satt_version": (_I, []),
    "satt_strerror": (C.c_char_p, [_I]),
    "satt_arch_supported": (_I, [_I]),
    "satt_gemm": (_I, [C.POINTER(GemmParams), _P]),
    "satt_gemm_path": (_I, [C.POINTER(GemmParams)]),
    "satt_gemm_ws_floats": (c_i64, [C.POINTER(GemmParams)]),
    "satt_shadow_pack": (_I, [_P, _P, _I, _P, _P, _P]),
    "satt_embedding_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "satt_embedding_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "satt_embedding_bwd_rows": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "satt_act_bwd": (_I, [_P, c_i64, _P, c_i64, _P, c_i64, _I, _I, _I, _F, _P]),
    "satt_act_bwd_res": (_I, [_P, c_i64, _P, c_i64, _P, c_i64, _P, c_i64, _I, _I, _I, _F, _P]),
    "satt_bn_ws_floats": (c_i64, [_I, _I]),
    "satt_bn_fwd": (_I, [_P, c_i64, _P, _P, _P, c_i64, _P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _P]),
    "satt_bn_infer": (_I, [_P, c_i64, _P, _P, _P, _P, _P, c_i64, _I, _I, _F, _I, _P]),
    "satt_bn_bwd": (_I, [_P, c_i64, _P, c_i64, _P, _P, _P, _P, _P, c_i64, _P, _P, _P, _I, _I, _I, _P]),
    "satt_bn_fused_ws_floats": (c_i64, [_I, _I]),
    "satt_bn_fused_sync_words": (_I, [_I]),
    "satt_bn_fwd_fused": (_I, [_P, c_i64, _P, _P, _P, c_i64, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _P]),
    "satt_bn_bwd_fused": (_I, [_P, c_i64, _P, c_i64, _P, _P, _P, _P, _P, c_i64, _P, _P, _P, _P, _I, _I, _I, _P]),
    "satt_maxpool_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "satt_bn_maxpool_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P]),
    "satt_maxpool_bn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "satt_maxpool_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "satt_highway_fwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "satt_highway_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "satt_highway_stack_fwd": (_I, [_P, C.POINTER(HighwayLayer), _I, _I, _I, _P]),
    "satt_highway_stack_bwd": (_I, [_P, _P, C.POINTER(HighwayLayer), _I, _I, _I, _P, _P]),
    "satt_colsum": (_I, [_P, c_i64, _P, _I, _I, _I, _P]),
    "satt_loc_filter_dw": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "satt_axpby": (_I, [_P, c_i64, _P, c_i64, _I, _I, _F, _F, _P]),
    "satt_seq_mask": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "satt_bcast_add": (_I, [_P, _P, _I, _I, _I, _P]),
    "satt_segment_colsum": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "satt_to_bf16": (_I, [_P, c_i64, _P, _I, _I, _I, _P]),
    "satt_softmax_fwd": (_I, [_P, _P, _P, _I, _I, _F, _I, c_u32, _F, c_u32, _P, _P]),
    "satt_softmax_bwd": (_I, [_P, _P, _P, _I, _I, _F, _I, c_u32, _F, c_u32, _P, _P]),
    "satt_small_attn_supported": (_I, [_I, _I]),
    "satt_small_attn_fwd": (_I, [_P, c_i64, _P, _P, c_i64, _I, _I, _I, _I, _F, C.c_uint32, _F, C.c_uint32, _P, _P]),
    "satt_small_attn_bwd": (_I, [_P, c_i64, _P, _P, c_i64, _P, c_i64, _P, _I, _I, _I, _I, _F, C.c_uint32, _F, C.c_uint32, _P, _P]),
    "satt_flash_attn_fwd": (_I, [_P, _P, _P, c_i64, _P, c_i64, _P, _I, _I, _I, _I, _F, _I, c_u32, _F, c_u32, _P, _P]),
    "satt_flash_attn_bwd": (_I, [_P, _P, _P, c_i64, _P, _P, c_i64, _P, _P, _P, _P, _P, c_i64, _I, _I, _I, _I, _F, _I, c_u32,
                                 _F, c_u32, _P, _P]),
    "satt_flash_attn_bwd_tiles": (_I, [_P, _P, _P, c_i64, _P, _P, c_i64, _P, _P, _P, _P, _P, c_i64, _I, _I, _I, _I, _F, _I, c_u32,
                                       _F, c_u32, _P, _I, _I, _I, _P]),
    "satt_flash_attn_fwd_b": (_I, [_P, _P, _P, c_i64, _P, c_i64, _P, _I, _I, _I, _I, _F, _I, c_u32, _F, c_u32, _P, _P, _P, _P, c_i64, _P]),
    "satt_flash_attn_bwd_tiles_b": (_I, [_P, _P, _P, c_i64, _P, _P, c_i64, _P, c_i64, _P, _P, _P, _P, _P, c_i64, _I, _I, _I, _I, _F,
                                         _I, c_u32, _F, c_u32, _P, _I, _I, _I, _P]),
    "satt_stream_probe": (_I, [_P, _P, C.c_uint, _P, _P]),
    "satt_debug_poison_lds": (_I, [C.c_uint, _I, _P]),
    "satt_softmax_rows": (_I, [_P, c_i64, _P, c_i64, _I, _I, C.c_float, _P]),
    "satt_dropout": (_I, [_P, c_i64, _P, c_i64, _I, _I, c_u32, _F, c_u32, _P, _P]),
    "satt_lstm_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, c_u32, c_u32, _P, C.POINTER(c_u32),
                           C.POINTER(c_u32), _P, c_i64, _P, _P, _P, _P, _P]),
    "satt_lstm_bwd": (_I, [_P, c_i64, _P, _P, _I, _I, _I, _I, _I, _F, _F, c_u32, c_u32, _P, C.POINTER(c_u32),
                           C.POINTER(c_u32), _P, _P, _P, _P, _P]),
    "satt_lstm_cluster_ws_bytes": (c_i64, [_I, _I, _I]),
    "satt_lstm_cluster_pack_elems": (c_i64, [_I]),
    "satt_lstm_cluster_pack": (_I, [_P, c_i64, _I, _I, _P, _P, _P]),
    "satt_lstm_cluster_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _F, c_u32, c_u32, _P, c_u32, c_u32, _P, c_i64,
                                   _P, _P, _P, _P, _P, _I, _I, _P]),
    "satt_lstm_cluster_pack_in_elems": (c_i64, [_I, _I]),
    "satt_lstm_cluster_pack_in": (_I, [_P, c_i64, _I, _I, _I, _P, _P]),
    "satt_lstm_cluster_fwd_x": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _F, c_u32, c_u32, _P, c_u32, c_u32, _P, c_i64,
                                     _P, _P, _P, _P, _P, _I, _I, _P, c_i64, _I, _P, _P, _P]),
    "satt_lstm_cluster_bwd": (_I, [_P, c_i64, _P, _I, _I, _I, _I, _I, _F, _F, c_u32, c_u32, _P, c_u32, c_u32, _P, _P,
                                   _P, _P, _P, _I, _I, _P, _P]),
    "satt_lstm_cluster_status": (_I, [_P, _I, _I, _I, _P]),
    "satt_lstm_cluster_fastpath": (_I, [_P, _I, _I, _I, _P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "satt_lstm_cluster_check": (_I, [_I, _I, _I, _I]),
    "satt_lstm_cluster_residency": (_I, [_I, _I, _I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "satt_attn_rnn_fwd": (_I, [C.POINTER(AttnRnnParams), _P]),
    "satt_attn_rnn_bwd": (_I, [C.POINTER(AttnRnnBwdParams), _P]),
    "satt_attn_param_grads": (_I, [C.POINTER(AttnRnnParams), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "satt_attn_param_grads_range": (_I, [C.POINTER(AttnRnnParams), _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "satt_attn_param_grads_acc_doubles": (c_i64, [C.POINTER(AttnRnnParams)]),
    "satt_attn_param_grads_acc": (_I, [C.POINTER(AttnRnnParams), _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "satt_attn_param_grads_finish": (_I, [C.POINTER(AttnRnnParams), _P, _P, _P, _P, _P, _P]),
    "satt_attn_cluster_ws_bytes": (c_i64, [C.POINTER(AttnRnnParams), _I]),
    "satt_attn_cluster_state_floats": (c_i64, [C.POINTER(AttnRnnParams), _I]),
    "satt_attn_cluster_pack_elems": (c_i64, [_I, _I, _I, _I]),
    "satt_attn_cluster_pack": (_I, [_P, c_i64, _P, _P, _I, _I, _I, _P]),
    "satt_attn_cluster_fold": (_I, [C.POINTER(AttnRnnParams), _I]),
    "satt_attn_cluster_fwd": (_I, [C.POINTER(AttnClusterParams), _P]),
    "satt_attn_cluster_bwd": (_I, [C.POINTER(AttnClusterBwdParams), _P]),
    "satt_attn_cluster_status": (_I, [C.POINTER(AttnRnnParams), _I, _P, _P]),
    "satt_attn_cluster_fastpath": (_I, [C.POINTER(AttnRnnParams), _I, _P, _P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "satt_attn_cluster_check": (_I, [C.POINTER(AttnRnnParams), _I]),
    "satt_attn_cluster_residency": (_I, [C.POINTER(AttnClusterParams), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "satt_attn_cluster_bwd_residency": (_I, [C.POINTER(AttnClusterBwdParams), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "satt_loss_fwd_bwd": (_I, [_P, c_i64, _P, _P, _P, c_i64, _P, _P, _I, _I, _I, _I, _I, _P, _P, c_i64, _P, c_i64,
                               _P, _P]),
    "satt_loss_fwd_bwd_presummed": (_I, [_P, c_i64, _P, _P, _P, c_i64, _P, _P, _I, _I, _I, _I, _I, _P, _P, c_i64, _P, c_i64,
                                         _P, _P]),
    "satt_loss_mask_sums": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "satt_dec_linear": (_I, [C.POINTER(DecLinearParams), _P]),
    "satt_dec_linear2": (_I, [C.POINTER(DecLinearParams), C.POINTER(DecLinearParams), _P]),
    "satt_dec_linear_chain": (_I, [C.POINTER(DecLinearParams), C.c_int, C.POINTER(DecLinearParams), _P]),
    "satt_dec_attention": (_I, [C.POINTER(DecAttentionParams), _P]),
    "satt_dec_mega_supported": (_I, [C.POINTER(DecMegaParams)]),
    "satt_dec_mega_scratch_floats": (c_i64, [_I, _I, _I]),
    "satt_dec_mega": (_I, [C.POINTER(DecMegaParams), _P]),
    "satt_dec_self_attn": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "satt_l2_reg": (_I, [_P, _P, _P, _I, _F, _P, _P, _P]),
    "satt_sumsq": (_I, [_P, c_i64, _P, _P]),
    "satt_sumsq_state_floats": (_I, []),
    "satt_adam_step": (_I, [_P, _P, _P, _P, c_i64, _P, _P, _P, _F, _I, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P]),
    "satt_poison_on_error": (_I, [_P, _P, _P, _P, _P]),
}

_lib = None


class SattError(RuntimeError):
    pass


def lib():
    """Load libsatt_hip.so (once).  Raises if it is missing: there is no CPU / eager fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SattError("libsatt_hip.so not found at %s — run `python __graft_entry__.py build` "
                            "(hipcc --offload-arch=gfx950); the product path has no fallback" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        pat = os.environ.get("SATT_DEBUG_POISON_LDS")
        _lib = _PoisonedLds(l, int(pat, 16)) if pat else l
    return _lib


class _PoisonedLds:
    """SATT_DEBUG_POISON_LDS=<hex pattern> (diagnostics): every entry point that takes a stream is preceded, on that stream, by a
    launch that leaves the pattern in every LDS word of every CU - a kernel that reads LDS before writing it, and whose result then
    depends on the pattern (7fc00000 = NaN against 3f800000 = 1.0), is found by running the tests both ways.  r6: the persistent
    decode kernel multiplied an unwritten LDS tail by zero weights (0 x NaN -> ReLU -> 0: DESIGN.md 3.5)."""

    def __init__(self, l, pattern):
        self._l, self._pattern = l, pattern

    def __getattr__(self, name):
        fn = getattr(self._l, name)
        sig = SIGNATURES.get(name)
        if sig is None or not sig[1] or sig[1][-1] is not _P or name in ("satt_debug_poison_lds", "satt_stream_probe"):
            return fn
        poison, pat = self._l.satt_debug_poison_lds, self._pattern

        def call(*a):
            poison(pat, 0, a[-1])
            return fn(*a)
        setattr(self, name, call)
        return call


def check(rc, what=""):
    if rc != 0:
        msg = lib().satt_strerror(rc).decode()
        raise SattError("%s failed: %s (%d)" % (what or "satt call", msg, rc))
