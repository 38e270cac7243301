"""CPU tests of the drop-in boundary (-m "not gpu"): the C-ABI library builds for gfx950, loads, and exports
every symbol include/satt_hip.h declares; the ctypes structs mirror the C structs; host-side logic (param
layout, hparams surface, LR schedule) is consistent.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import satt_amd  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "satt_hip.h")


def header_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(satt_[a-z0-9_]+)\s*\(", txt)))


def test_build_and_exports():
    import __graft_entry__ as ge
    so = ge.build()
    assert os.path.exists(so)
    lib = ctypes.CDLL(so)
    syms = header_symbols()
    assert len(syms) >= 28
    for s in syms:
        assert hasattr(lib, s), "missing export " + s
    from satt_amd import _lib
    assert sorted(_lib.SIGNATURES) == syms, set(syms) ^ set(_lib.SIGNATURES)
    lib.satt_version.restype = ctypes.c_int
    assert lib.satt_version() >= 100
    lib.satt_strerror.restype = ctypes.c_char_p
    assert b"bad argument" in lib.satt_strerror(-1)


def test_struct_layouts_match_c():
    """sizeof / offsetof of the ctypes mirrors == the C structs (compiled with the host compiler)."""
    from satt_amd import _lib
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "satt_hip.h"
int main() {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(satt_gemm_params), offsetof(satt_gemm_params, precision),
         offsetof(satt_gemm_params, bias), sizeof(satt_attn_rnn_params), offsetof(satt_attn_rnn_params, hstate),
         sizeof(satt_attn_rnn_bwd_params), offsetof(satt_attn_rnn_bwd_params, dfl));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", offsetof(satt_gemm_params, ws), offsetof(satt_attn_rnn_params, acum),
         sizeof(satt_attn_cluster_params), sizeof(satt_attn_cluster_bwd_params), sizeof(satt_dec_linear_params),
         offsetof(satt_dec_linear_params, min_steps), offsetof(satt_dec_linear_params, lstm_H),
         sizeof(satt_dec_attention_params), offsetof(satt_dec_attention_params, step));
  printf("%zu %zu %zu %zu %zu\n", sizeof(satt_dec_mega_params), offsetof(satt_dec_mega_params, zc), offsetof(satt_dec_mega_params, Wp0),
         offsetof(satt_dec_mega_params, step), offsetof(satt_dec_mega_params, nsteps));
  return 0; }'''
    d = "/tmp/satt_struct_test"
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "t.c"), "w").write(src)
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
    vals = list(map(int, subprocess.check_output([os.path.join(d, "t")]).split()))
    G, A, Bp = _lib.GemmParams, _lib.AttnRnnParams, _lib.AttnRnnBwdParams
    DL, DA = _lib.DecLinearParams, _lib.DecAttentionParams
    assert vals == [ctypes.sizeof(G), G.precision.offset, G.bias.offset, ctypes.sizeof(A), A.hstate.offset,
                    ctypes.sizeof(Bp), Bp.dfl.offset,
                    G.ws.offset, A.acum.offset, ctypes.sizeof(_lib.AttnClusterParams), ctypes.sizeof(_lib.AttnClusterBwdParams),
                    ctypes.sizeof(DL), DL.min_steps.offset, DL.lstm_H.offset, ctypes.sizeof(DA), DA.step.offset,
                    ctypes.sizeof(_lib.DecMegaParams), _lib.DecMegaParams.zc.offset, _lib.DecMegaParams.Wp0.offset,
                    _lib.DecMegaParams.step.offset, _lib.DecMegaParams.nsteps.offset]


def test_missing_library_fails_loudly(monkeypatch):
    from satt_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsatt_hip.so")
    with pytest.raises(_lib.SattError):
        _lib.lib()


def test_param_layout_matches_survey():
    from satt_amd.params import ModelConfig, layout, param_shapes
    from oracle import torch_ref
    cfg = ModelConfig()
    assert sum(int(np.prod(s)) for _, s in param_shapes(cfg)) == 6246104
    assert param_shapes(cfg) == torch_ref.param_shapes(torch_ref.Cfg())
    lay, n = layout(cfg)
    assert all(o % 8 == 0 for o, _ in lay.values()) and n >= 6246104      # bf16 shadows at the same offsets: 16-byte aligned
    offs = [lay[k][0] for k, _ in param_shapes(cfg)]
    assert offs == sorted(offs)
    assert lay["dec.prenet0.W"][0] > lay["enc.sa.t.b"][0]                      # encoder bucket precedes decoder


def test_hparams_surface():
    from satt_amd.hparams import hparams, hparams_debug_string
    hp = hparams.copy()
    assert hp.suffle_buffer_size == 64 and hp.tacotron_model == "ExtendedTacotronV1Model"   # reference defaults
    hp.parse_json(open(os.path.join(ROOT, "examples", "ljspeech", "self-attention-tacotron.json")).read())
    assert hp.tacotron_model == "DualSourceSelfAttentionTacotronModel" and hp.attention == "forward"
    assert hp.attention_kernel == 10 and hp.attention_filters == 5 and hp.initial_learning_rate == 0.0005
    hp.parse("batch_size=16,attention2=additive,use_postnet_v2=False,encoder_prenet_out_units=[128,64]")
    assert hp.batch_size == 16 and hp.encoder_prenet_out_units == [128, 64] and hp.use_postnet_v2 is False
    with pytest.raises(ValueError):
        hp.parse("no_such_key=1")
    assert "batch_size: 16" in hparams_debug_string(hp)
    from satt_amd.params import ModelConfig
    c = ModelConfig.from_hparams(hp)
    assert c.att1_units == 224 and c.att_kernel == 10 and c.r == 2


def test_lr_schedule():
    from oracle import torch_ref
    assert abs(torch_ref.learning_rate(5e-4, 3999) - 5e-4) < 1e-12       # peaks at lr0 when s = 4000
    assert torch_ref.learning_rate(5e-4, 0) < torch_ref.learning_rate(5e-4, 100)


@pytest.mark.parametrize("growth", [1.4, 2.0])
@pytest.mark.parametrize("Td,NC,tail", [(400, 6, (3, 4)), (400, 8, (4, 8)), (37, 6, (3, 4)), (1000, 7, (3, 4)), (5, 6, (3, 4)),
                                        (400, 8, (6, 3)), (250, 8, (6, 3)), (1000, 8, (6, 3)), (17, 8, (6, 3))])
def test_layer_pipeline_schedule(Td, NC, tail, growth):
    """Host logic of the recurrent layer pipeline: chunk bounds tile [0, Td); the pieces of the single-launch backward
    attention kernel tile it in processing order (late to early), never cross a chunk, stay within the 16 counters of
    the C-ABI struct, and the per-chunk `ready` values are the running piece counts; merging only touches the leading
    (latest) entries."""
    from satt_amd.engine import Engine

    class E:       # the schedule helpers only read these attributes
        pipeline_tail = tail
        pipeline_growth = growth
    bounds = Engine._chunk_bounds(E, Td, NC)
    assert bounds[0][0] == 0 and bounds[-1][1] == Td
    assert all(a1 == b0 for (_, a1), (b0, _) in zip(bounds[:-1], bounds[1:])) and all(b > a for a, b in bounds)
    pieces, upto = Engine._backward_pieces(bounds, Td)
    assert len(pieces) <= 16 and len(upto) == len(bounds) and upto[-1] == len(pieces)
    assert pieces[0][1] == Td and pieces[-1][0] == 0
    assert all(p0 == q1 for (p0, _), (_, q1) in zip(pieces[:-1], pieces[1:])) and all(b > a for a, b in pieces)
    lo = 0
    for (b0, b1), hi in zip(reversed(bounds), upto):          # the pieces of a chunk lie inside it
        assert hi > lo and pieces[lo][1] == b1 and pieces[hi - 1][0] == b0
        lo = hi
    chunks = [(p0, p1, r) for r, (p0, p1) in enumerate(pieces)]
    merged = Engine._merge_leading(chunks, Td)
    assert merged[0][1] == Td and merged[-1][0] == 0 and [m[2] for m in merged] == sorted(m[2] for m in merged)
    assert all(m0 == n1 for (m0, _, _), (_, n1, _) in zip(merged[:-1], merged[1:]))
    k = len(chunks) - len(merged)                              # entries folded into the first one
    assert merged[0] == (chunks[k][0], Td, k) and merged[1:] == chunks[k + 1:]
    assert k == 0 or merged[0][1] - merged[0][0] <= max(1, (3 * Td) // 10)


def test_chunk_bounds_honour_an_explicit_tail():
    """the `tail` argument (pipeline_tail_fwd / SATT_TAIL_FWD) used to be shadowed by a local list and ignored"""
    from satt_amd.engine import Engine

    class E:
        pipeline_tail = (6, 3)
        pipeline_growth = 1.4
    default = Engine._chunk_bounds(E, 400, 8)
    assert Engine._chunk_bounds(E, 400, 8, tail=(6, 3)) == default
    other = Engine._chunk_bounds(E, 400, 8, tail=(3, 8))
    assert other != default and other[0][0] == 0 and other[-1][1] == 400
    assert other[-1][1] - other[-1][0] == max(8, 400 // (8 * 8))          # smallest tail chunk = Td / (tdiv * NC), at least 8 steps


@pytest.mark.parametrize("Td", [5, 40, 64, 100, 128, 250, 400, 401, 1000])
@pytest.mark.parametrize("nsuf", [1, 2, 3])
def test_split_head_row(Td, nsuf):
    """Host logic of the split decoder head: the suffix starts on a 64-row tile boundary at or below the start of the nsuf-th last
    chunk, so the chunks processed first lie entirely inside the suffix and every other chunk can be told apart by its start."""
    from satt_amd.engine import Engine

    class E:
        pipeline_tail = Engine.pipeline_tail
        pipeline_growth = Engine.pipeline_growth
    bounds = Engine._chunk_bounds(E, Td, 8)
    t_a = Engine._head_split_row(bounds, nsuf, 64)
    assert t_a % 64 == 0 and 0 <= t_a <= bounds[-1][0]
    first = list(reversed(bounds))[:min(nsuf, len(bounds))]
    assert all(b0 >= t_a for b0, _ in first)                     # the first chunks only need the suffix rows
    assert t_a + 64 > bounds[-min(len(bounds), nsuf)][0]         # and the suffix is no larger than a tile needs it to be
    if t_a == 0:
        assert bounds[-min(len(bounds), nsuf)][0] < 64           # no split below one tile

