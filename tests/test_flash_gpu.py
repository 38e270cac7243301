"""Fused self-attention (csrc/flash.hip) against an fp64 PyTorch restatement of ScaledDotProductAttentionMechanism
(reference modules/self_attention.py:45-65,79-86): softmax(q k^T / sqrt(hd) [+ causal mask]) -> dropout on the
probabilities (counter-based mask, same counter as satt_softmax_fwd / oracle/rng.py) -> . v, forward and all three input
gradients.  Operands are bf16-rounded on both sides; the kernel additionally rounds P and dS to bf16 for the second
products, so the bar is 2e-2 of the largest reference magnitude (forward typically 3e-3)."""
import math

import numpy as np
import pytest
import torch

import satt_amd  # noqa: F401
from oracle import rng

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ref_attention(kvq, B, T, H, hd, causal, keep, dscale):
    D = H * hd
    x = kvq.double().view(B, T, 3, H, hd).bfloat16().double()            # bf16 operands, as the MFMA sees them
    x.requires_grad_(True)
    k, v, q = (x[:, :, i].transpose(1, 2) for i in range(3))             # [B, H, T, hd]
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), 1), float("-inf"))
    p = torch.softmax(s, -1)
    pd = p * keep * dscale if keep is not None else p
    o = (pd @ v).transpose(1, 2).reshape(B * T, D)
    return x, o


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("rate", [0.0, 0.05])
@pytest.mark.parametrize("B,T,H", [(2, 70, 2), (2, 400, 2), (3, 64, 1), (1, 130, 3)])
def test_flash_attention_fwd_bwd(B, T, H, causal, rate):
    from satt_amd import ops
    ops.set_precision("bf16")
    hd, D = 128, H * 128
    g = torch.Generator().manual_seed(B * 1000 + T + H)
    kvq = torch.randn(B * T, 3 * D, generator=g)
    dout = torch.randn(B * T, D, generator=g)
    seedv = 321
    keep = None
    drop = ops.Drop(rate, 16, torch.tensor([seedv], dtype=torch.int32, device=DEV))
    if rate > 0:
        keep = torch.from_numpy(rng.keep_mask(seedv, 16, (B * H * T, T), rate)).double().view(B, H, T, T)
    x, o_ref = ref_attention(kvq, B, T, H, hd, causal, keep, drop.scale)
    o_ref.backward(dout.double())
    dkvq_ref = x.grad.reshape(B * T, 3 * D)
    kd = kvq.to(DEV).contiguous()
    o = torch.zeros(B * T, D, device=DEV)
    lse = torch.zeros(B * H, T, device=DEV)
    ops.flash_attn_fwd(kd, D, o, lse, B, T, H, 1.0 / math.sqrt(hd), causal, drop)
    err = float((o.double().cpu() - o_ref.detach()).abs().max() / o_ref.detach().abs().max())
    print("forward rel err %.3e" % err)
    assert err < 2e-2
    dkvq = torch.full((B * T, 3 * D), float("nan"), device=DEV)      # every element must be written
    delta = torch.empty(B * H, T, device=DEV)
    ops.flash_attn_bwd(kd, D, o, dout.to(DEV).contiguous(), lse, delta, dkvq, B, T, H, 1.0 / math.sqrt(hd), causal, drop)
    assert bool(torch.isfinite(dkvq).all())
    for name, sl in (("dK", slice(0, D)), ("dV", slice(D, 2 * D)), ("dQ", slice(2 * D, 3 * D))):
        a, b = dkvq[:, sl].double().cpu(), dkvq_ref[:, sl]
        e = float((a - b).abs().max() / b.abs().max())
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        print("%s rel err %.3e cos %.6f" % (name, e, cos))
        assert e < 2e-2 and cos > 0.9999, (name, e, cos)


def test_flash_matches_the_unfused_path_in_the_engine():
    """decoder self-attention block of the engine: fused kernels (benchmark precision) vs the GEMM + softmax path with
    the same dropout masks - loss and every gradient of one LJSpeech-dims train step"""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    ops.set_precision("bf16")
    batch = synthetic_batch(4, 40, 128, seed=3, min_source_length=20, min_target_steps=30)
    res = {}
    for fused in (True, False):
        eng = Engine(ModelConfig(), "cuda", param_seed=1, rng_seed=9)
        b = eng.to_device_batch(batch)
        if not fused:
            orig = ops.flash_attn_supported
            ops.flash_attn_supported = lambda hd: False
        try:
            eng.zero_grad()
            ctx = eng.forward(b, True)
            eng.backward(ctx)
            torch.cuda.synchronize()
        finally:
            if not fused:
                ops.flash_attn_supported = orig
        assert (ctx["dec_mha"]["lse"] is not None) == fused
        res[fused] = (float(eng.losses[2]), eng.grad.detach().cpu().numpy().astype(np.float64))
    assert abs(res[True][0] - res[False][0]) < 2e-3
    ga, gb = res[True][1], res[False][1]
    cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    print("loss %.6f vs %.6f, gradient cosine %.6f" % (res[True][0], res[False][0], cos))
    assert cos > 0.999


@pytest.mark.parametrize("rate", [0.0, 0.05])
@pytest.mark.parametrize("B,T,H,split", [(2, 400, 2, 6), (2, 250, 2, 3), (1, 130, 3, 1), (3, 64 * 3, 1, 2)])
def test_flash_backward_suffix_prefix_split_is_the_single_launch(B, T, H, split, rate):
    """satt_flash_attn_bwd_tiles: a causal backward as a launch over the tile suffix [split, nt) and one over the prefix
    [0, split) writes bit for bit what the single launch writes, and the suffix rows are final after the FIRST launch (the
    training step's backward pipeline starts on them while the prefix launch is still running)."""
    from satt_amd import ops
    ops.set_precision("bf16")
    hd, D = 128, H * 128
    g = torch.Generator().manual_seed(7 * T + H)
    kd = torch.randn(B * T, 3 * D, generator=g).to(DEV)
    dout = torch.randn(B * T, D, generator=g).to(DEV)
    drop = ops.Drop(rate, 16, torch.tensor([99], dtype=torch.int32, device=DEV))
    o, lse = torch.zeros(B * T, D, device=DEV), torch.zeros(B * H, T, device=DEV)
    sc = 1.0 / math.sqrt(hd)
    ops.flash_attn_fwd(kd, D, o, lse, B, T, H, sc, True, drop)
    whole = torch.full((B * T, 3 * D), float("nan"), device=DEV)
    ops.flash_attn_bwd(kd, D, o, dout, lse, torch.empty(B * H, T, device=DEV), whole, B, T, H, sc, True, drop)
    parts = torch.full((B * T, 3 * D), float("nan"), device=DEV)
    delta = torch.empty(B * H, T, device=DEV)
    nt = (T + ops.FLASH_TILE - 1) // ops.FLASH_TILE
    ops.flash_attn_bwd(kd, D, o, dout, lse, delta, parts, B, T, H, sc, True, drop, tiles=(split, nt))
    torch.cuda.synchronize()
    rows = parts.view(B, T, 3 * D)
    t_a = split * ops.FLASH_TILE
    assert torch.equal(rows[:, t_a:], whole.view(B, T, 3 * D)[:, t_a:])            # suffix rows final
    assert bool(torch.isnan(rows[:, :t_a]).all())                                  # nothing below touched
    ops.flash_attn_bwd(kd, D, o, dout, lse, delta, parts, B, T, H, sc, True, drop, tiles=(0, split))
    torch.cuda.synchronize()
    assert torch.equal(parts, whole)
    # a proper sub-range of a non-causal problem is refused
    with pytest.raises(Exception):
        ops.flash_attn_bwd(kd, D, o, dout, lse, delta, parts, B, T, H, sc, False, drop, tiles=(1, nt))


@pytest.mark.parametrize("B,Ti,Tm", [(4, 40, 256), (32, 60, 500)])
def test_split_head_backward_equals_the_unsplit_step(B, Ti, Tm):
    """Engine.head_split: the decoder self-attention backward as suffix + prefix launches with the recurrent pipeline started
    behind the suffix - same loss and gradients as the single-launch head (LJSpeech dimensions, bf16)"""
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    ops.set_precision("bf16")
    batch = synthetic_batch(B, Ti, Tm, seed=5, min_source_length=20, min_target_steps=Tm // 4)
    res = {}
    for split in (True, False):
        eng = Engine(ModelConfig(), "cuda", param_seed=1, rng_seed=9)
        eng.head_split = split
        b = eng.to_device_batch(batch)
        for _ in range(2):          # the second pass is the steady state (streams, packs, shadows warm)
            eng.zero_grad()
            ctx = eng.forward(b, True)
            eng.backward(ctx)
            torch.cuda.synchronize()
            eng.check_clusters(ctx)
        assert (eng._head_split is not None) == (split and ctx["chunks"] > 1)
        res[split] = (float(eng.losses[2]), eng.grad.detach().clone())
    print("loss %.9f (split) vs %.9f" % (res[True][0], res[False][0]))
    assert abs(res[True][0] - res[False][0]) < 1e-5 * abs(res[False][0])
    # (the split changes no arithmetic; the weight-gradient GEMMs' split-K reductions are what differs between any two runs)
    d = float((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max())
    print("max |gradient difference| / max |gradient|, split vs unsplit: %.3e" % d)
    assert d < 1e-5, d


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("rate", [0.0, 0.05])
@pytest.mark.parametrize("B,T,H", [(2, 400, 2), (2, 70, 2), (1, 130, 3), (3, 64, 1)])
def test_flash_bf16_copies_and_bf16_source_backward(B, T, H, causal, rate):
    """satt_flash_attn_fwd_b / satt_flash_attn_bwd_tiles_b: the forward kernel's bf16 copies of K | V | Q are the nearest-even
    roundings of the fp32 rows (every row written, outputs unchanged), and the backward that reads them (and the bf16 copy of d o
    its delta pass writes) returns bit for bit what the fp32-source backward returns."""
    from satt_amd import ops
    ops.set_precision("bf16")
    hd, D = 128, H * 128
    g = torch.Generator().manual_seed(11 * T + H)
    kd = torch.randn(B * T, 3 * D, generator=g).to(DEV)
    dout = torch.randn(B * T, D, generator=g).to(DEV)
    drop = ops.Drop(rate, 16, torch.tensor([5], dtype=torch.int32, device=DEV))
    sc = 1.0 / math.sqrt(hd)
    o0, lse0 = torch.zeros(B * T, D, device=DEV), torch.zeros(B * H, T, device=DEV)
    ops.flash_attn_fwd(kd, D, o0, lse0, B, T, H, sc, causal, drop)
    o1, lse1 = torch.zeros(B * T, D, device=DEV), torch.zeros(B * H, T, device=DEV)
    kb = torch.full((B * T, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.flash_attn_fwd(kd, D, o1, lse1, B, T, H, sc, causal, drop, kvq_b=kb)
    torch.cuda.synchronize()
    assert torch.equal(o0, o1) and torch.equal(lse0, lse1)
    assert torch.equal(kb, kd.bfloat16())                     # torch rounds to nearest-even as well
    ref = torch.full((B * T, 3 * D), float("nan"), device=DEV)
    ops.flash_attn_bwd(kd, D, o0, dout, lse0, torch.empty(B * H, T, device=DEV), ref, B, T, H, sc, causal, drop)
    got = torch.full((B * T, 3 * D), float("nan"), device=DEV)
    dob = torch.full((B * T, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.flash_attn_bwd(None, D, o0, dout, lse0, torch.empty(B * H, T, device=DEV), got, B, T, H, sc, causal, drop, kvq_b=kb, do_b=dob)
    torch.cuda.synchronize()
    assert torch.equal(dob, dout.bfloat16())
    assert torch.equal(got, ref)
    if causal and T > 64:       # tile ranges of the bf16-source form
        nt = (T + ops.FLASH_TILE - 1) // ops.FLASH_TILE
        parts = torch.full((B * T, 3 * D), float("nan"), device=DEV)
        delta = torch.empty(B * H, T, device=DEV)
        ops.flash_attn_bwd(None, D, o0, dout, lse0, delta, parts, B, T, H, sc, True, drop, tiles=(nt - 1, nt), kvq_b=kb, do_b=dob)
        ops.flash_attn_bwd(None, D, o0, dout, lse0, delta, parts, B, T, H, sc, True, drop, tiles=(0, nt - 1),
                           kvq_b=kb, do_b=dob)
        torch.cuda.synchronize()
        assert torch.equal(parts, ref)
