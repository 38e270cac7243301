"""The reference's module CALL contracts on the GPU: encoder_factory(...) / decoder_factory(...)
return callables - `encoder(inputs, input_lengths) -> (lstm_output, self_attention_output, alignments)` (reference
modules/module.py:425-441), `decoder((src1, src2), attention1_fn=..., ...) -> (mel, stop_token, state)` (:1493-1559) - and the
mechanisms `attention_fn(memory, memory_sequence_length)` returns expose their alignment histories.  Judged by the float64 oracle
(oracle/torch_ref.py encoder / decoder) in the exact-fp32 mode, LJSpeech dimensions, dropout / zoneout on."""
import json
import os

import numpy as np
import pytest
import torch

from common import make_params, oracle_cfg, rel_err, small_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hp():
    from satt_amd.hparams import hparams as default_hparams
    hp = default_hparams.copy()
    d = json.load(open(os.path.join(ROOT, "examples", "ljspeech", "self-attention-tacotron.json"))); d.pop("_comment", None)
    hp.parse_json(json.dumps(d))
    return hp


@pytest.fixture()
def setup():
    from oracle import torch_ref
    from satt_amd import ops
    from satt_amd.engine import Engine
    ops.set_precision("f32")
    cfg, P = make_params(dict(), seed=1)
    batch = small_batch(cfg, 4, 21, 24, seed=3)
    eng = Engine(cfg, "cuda", params=P, rng_seed=7)
    Pt = torch_ref.to_torch(P)
    bt = torch_ref.batch_to_torch(batch)
    yield eng, P, Pt, batch, bt, oracle_cfg(dict())
    ops.set_precision("bf16")


def test_encoder_call_contract(setup):
    from oracle import torch_ref
    from satt_amd.models.models import encoder_factory
    eng, P, Pt, batch, bt, ocfg = setup
    enc = encoder_factory(_hp(), True).bind(eng)
    assert enc.name == "SelfAttentionCBHGEncoder" and enc.cbhg_out_units == 256            # still the descriptor it always was
    embedded = P["embedding"][batch["source"]]                                              # models/models.py:351
    lstm_out, sa_out, aligns = enc(embedded, input_lengths=batch["source_length"])
    col = {}
    r_lstm, r_sa, r_al = torch_ref.encoder(bt["source"], bt["source_length"], Pt, ocfg, True, 7, collect=col)
    assert tuple(lstm_out.shape) == (4, 21, 256) and tuple(sa_out.shape) == (4, 21, 32) and len(aligns) == 2
    errs = dict(lstm=rel_err(lstm_out.cpu().numpy(), r_lstm.detach().numpy()), sa=rel_err(sa_out.cpu().numpy(), r_sa.detach().numpy()),
                al=max(rel_err(aligns[h].cpu().numpy(), r_al.detach().numpy()[:, h]) for h in range(2)))
    print(errs)
    assert max(errs.values()) < 2e-4, errs
    with pytest.raises(ValueError):
        enc(embedded[:, :, :7])
    # an encoder that was never bound builds its own engine from the hparams (fresh parameters, like a layer's first call)
    fresh = encoder_factory(_hp(), False)
    out = fresh(embedded)
    assert tuple(out[0].shape) == (4, 21, 256) and fresh.engine is not eng and bool(torch.isfinite(out[0]).all())


def test_decoder_call_contract_teacher_fed_and_free_running(setup):
    from oracle import torch_ref
    from satt_amd.inference import infer
    from satt_amd.models.attention_factories import dual_source_attention_factory
    from satt_amd.models.models import decoder_factory
    eng, P, Pt, batch, bt, ocfg = setup
    hp = _hp()
    dec = decoder_factory(hp).bind(eng)
    a1, a2 = dual_source_attention_factory(hp)
    r_lstm, r_sa, _ = torch_ref.encoder(bt["source"], bt["source_length"], Pt, ocfg, True, 7)
    src = (r_lstm.detach().float(), r_sa.detach().float())
    mel, stop, state = dec(src, attention1_fn=a1, attention2_fn=a2, speaker_embed=None, is_training=True, is_validation=False,
                           teacher_forcing=False, memory_sequence_length=batch["source_length"],
                           memory2_sequence_length=batch["source_length"], target_sequence_length=batch["target_length"],
                           target=batch["mel"], teacher_alignments=(None, None), apply_dropout_on_inference=False)
    r_mel, r_stop, r_a1, r_a2, _ = torch_ref.decoder(r_lstm, r_sa, bt["source_length"], bt["mel"], Pt, ocfg, True, 7)
    errs = dict(mel=rel_err(mel.cpu().numpy(), r_mel.detach().numpy()), stop=rel_err(stop.cpu().numpy(), r_stop.detach().numpy()),
                a1=rel_err(state["alignments"][0].cpu().numpy(), r_a1.detach().numpy()),
                a2=rel_err(state["alignments"][1].cpu().numpy(), r_a2.detach().numpy()))
    print(errs)
    assert max(errs.values()) < 2e-4, errs
    # the mechanisms the attention_fns built for this call: kind, memory, and the alignment history of the call
    m1, m2 = state["attention_mechanisms"]
    assert (m1.kind, m2.kind) == ("forward", "additive") and m1.memory.shape == (4, 21, 256) and m2.memory.shape == (4, 21, 32)
    assert m1.alignments is state["alignments"][0] and tuple(m2.alignments.shape) == (4, 12, 21)
    assert np.allclose(m1.alignments.sum(-1).cpu().numpy(), 1.0, atol=1e-4)
    with pytest.raises(Exception):
        dec(src, attention1_fn=a1, attention2_fn=a2, is_training=True, target=None)
    with pytest.raises(Exception):
        eng.backward(dec.last_ctx)                        # a decoder-only context has no encoder state
    # free running (PREDICT: not training, no teacher forcing) == inference.infer on the same memories
    ctx = {"training": False, "batch": {}}
    b = eng.to_device_batch(batch)
    e_lstm, e_sa = eng._encode(b, False, ctx)
    mem = (e_lstm.view(4, 21, -1).clone(), e_sa.view(4, 21, -1).clone())
    mel_f, stop_f, st_f = dec(mem, attention1_fn=a1, attention2_fn=a2, is_training=False, is_validation=False,
                              memory_sequence_length=batch["source_length"])
    ref = infer(eng, b["source"], b["source_length"], max_steps=dec.max_iters)
    assert mel_f.shape == ref["mel"].shape
    assert rel_err(mel_f.cpu().numpy(), ref["mel"].cpu().numpy()) < 1e-5
    assert rel_err(st_f["alignments"][0].cpu().numpy(), ref["alignment1"].cpu().numpy()) < 1e-5
