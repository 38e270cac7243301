"""Large-tile bf16 GEMM families (csrc/gemm_tile.hip) against fp64 PyTorch on the same bf16-rounded operands:
forward Dense / Conv1D / conv bank and their input gradients through the bf16 weight shadows (satt_shadow_pack),
weight gradients with the fused bias column sum.  Every case asserts (satt_gemm_path) that the large-tile kernel - not
the generic one - produced the numbers.  Tolerance: 1e-3 relative to the largest output magnitude (fp32 accumulation
order is the only difference once the operands are rounded identically)."""
import math

import numpy as np
import pytest
import torch

import satt_amd  # noqa: F401
from oracle import rng, torch_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3


def T(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32, device=DEV).contiguous()


def bf(x):
    return x.bfloat16().float()


def close(a, b, tol, what=""):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
    print("%-28s rel_err=%.3e" % (what, err))
    assert err < tol, (what, err)


def make_weight(Wd):
    """ops.Weight for a standalone device tensor [K, N] or [taps, Cin, Cout]: shadows via satt_shadow_pack"""
    from satt_amd import ops
    shp = tuple(Wd.shape)
    taps, R, Cc = (1,) + shp if len(shp) == 2 else shp
    n = Wd.numel()
    st = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    sn = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    table = torch.tensor([0, taps, R, Cc], dtype=torch.int64, device=DEV)
    ops.shadow_pack(Wd, table, 1, st, sn)
    tshape = (Cc, R) if len(shp) == 2 else (taps, Cc, R)
    return ops.Weight(Wd, st.view(tshape), sn.view(shp))


class paths:
    """records satt_gemm_path of every GEMM issued inside the block"""

    def __enter__(self):
        from satt_amd import ops
        ops.gemm_path_log = self.log = []
        return self.log

    def __exit__(self, *exc):
        from satt_amd import ops
        ops.gemm_path_log = None


def test_shadow_pack_orientations():
    from satt_amd import ops
    g = torch.Generator().manual_seed(1)
    flat = torch.randn(40000, generator=g)
    # three weights at 8-aligned offsets: Dense [70, 33], conv [3, 40, 24], Dense [1, 17]
    entries = [(0, (70, 33)), (2400, (3, 40, 24)), (6000, (1, 17))]
    tab = []
    for off, shp in entries:
        taps, R, Cc = (1,) + shp if len(shp) == 2 else shp
        tab += [off, taps, R, Cc]
    fd = T(flat)
    st = torch.zeros(40000, dtype=torch.bfloat16, device=DEV); sn = torch.zeros_like(st)
    ops.shadow_pack(fd, torch.tensor(tab, dtype=torch.int64, device=DEV), len(entries), st, sn)
    for off, shp in entries:
        n = int(np.prod(shp))
        w = flat[off:off + n].view(shp)
        assert torch.equal(sn[off:off + n].float().cpu().view(shp), bf(w))
        wt = w.transpose(-1, -2).contiguous()
        assert torch.equal(st[off:off + n].float().cpu().view(wt.shape), bf(wt))
    assert float(sn[7000:].float().abs().max()) == 0        # nothing outside the listed tensors is touched


@pytest.mark.parametrize("M,N,K", [(256, 128, 544), (300, 130, 1000), (1000, 512, 40), (129, 64, 200), (2048, 1024, 136),
                                   (70, 161, 64), (12800, 256, 160), (5, 1024, 264), (640, 136, 8)])
def test_tile_linear_fwd_dx(M, N, K):
    from satt_amd import ops
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g); W = torch.randn(K, N, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    xd, Wd, bd, rd = T(x), T(W), T(b), T(res)
    Ww = make_weight(Wd)
    xr, Wr = bf(x).double(), bf(W).double()
    out = torch.empty(M, N, device=DEV)
    with paths() as log:
        ops.linear(xd, Ww, bd, out, act=ops.ACT_TANH, residual=rd)
    assert log == [1], log
    close(out, torch.tanh(xr @ Wr + b.double()) + res.double(), TOL, "linear tanh+res")
    # accumulate + relu/dropout epilogue (mask index = row * N + col)
    seed = torch.tensor([77], dtype=torch.int32, device=DEV)
    out2 = torch.empty(M, N, device=DEV)
    ops.linear(xd, Ww, bd, out2, act=ops.ACT_RELU, drop=ops.Drop(0.5, 5, seed))
    mask = torch.from_numpy(rng.keep_mask(77, 5, (M, N), 0.5))
    close(out2, torch.relu(xr @ Wr + b.double()) * mask * 2.0, TOL, "relu+dropout")
    ops.linear(xd, Ww, None, out2, accumulate=True)
    close(out2, torch.relu(xr @ Wr + b.double()) * mask * 2.0 + xr @ Wr, TOL, "accumulate")
    if N % 8 == 0:           # the dX shadow needs 16-byte aligned rows of W [K, N]
        dy = torch.randn(M, N, generator=g)
        dx = torch.empty(M, K, device=DEV)
        with paths() as log:
            ops.linear_dx(T(dy), Ww, dx)
        assert log == [1], log
        close(dx, bf(dy).double() @ Wr.T, TOL, "linear_dx")


def test_tile_linear_row_slices_and_row_batches():
    """W[r0:r1] of a larger matrix (the hoisted LSTM input projections) and per-sample row ranges (pipeline chunks)"""
    from satt_amd import ops
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(3)
    B, Tn, Kfull, K, N = 4, 50, 96 + 64, 96, 256
    x = torch.randn(B * Tn, K, generator=g); W = torch.randn(Kfull, N, generator=g) / 10
    xd, Wd = T(x), T(W)
    Ww = make_weight(Wd).rows(0, K)
    out = torch.zeros(B * Tn, N, device=DEV)
    bias = T(torch.randn(N, generator=g))
    with paths() as log:
        ops.linear_rows(xd, Ww, bias, out, B, Tn, 10, 37)
    assert log == [1], log
    ref = (bf(x).double() @ bf(W[:K]).double() + bias.double().cpu()).view(B, Tn, N)
    o3 = out.view(B, Tn, N)
    close(o3[:, 10:37], ref[:, 10:37], TOL, "linear_rows")
    assert float(o3[:, :10].abs().max()) == 0 and float(o3[:, 37:].abs().max()) == 0
    dy = torch.randn(B * Tn, N, generator=g)
    dx = torch.zeros(B * Tn, K, device=DEV)
    with paths() as log:
        ops.linear_dx_rows(T(dy), Ww, dx, B, Tn, 10, 37)
    assert log == [1], log
    refdx = (bf(dy).double() @ bf(W[:K]).double().T).view(B, Tn, K)
    close(dx.view(B, Tn, K)[:, 10:37], refdx[:, 10:37], TOL, "linear_dx_rows")
    assert float(dx.view(B, Tn, K)[:, 37:].abs().max()) == 0
    # second slice of the same matrix
    Wb = make_weight(Wd).rows(K, Kfull)
    xb = torch.randn(B * Tn, Kfull - K, generator=g)
    ob = torch.empty(B * Tn, N, device=DEV)
    with paths() as log:
        ops.linear(T(xb), Wb, None, ob)
    assert log == [1], log
    close(ob, bf(xb).double() @ bf(W[K:]).double(), TOL, "row slice [K:]")


@pytest.mark.parametrize("rows", [1, 5, 16, 17, 32])
@pytest.mark.parametrize("K,N", [(1024, 256), (768, 256), (1024, 544), (544, 1024), (1032, 100)])
def test_rows_gemm_few_rows_per_batch(rows, K, N):
    """gemm_rows_k (csrc/gemm_tile.hip): per-sample row ranges of <= 32 rows against a long reduction - the per-chunk
    products of the recurrent pipelines - with the reduction split over the four waves of a workgroup.  Forward form with
    bias, input-gradient form, and accumulation into previous contents; rows outside the range stay untouched."""
    from satt_amd import ops
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(rows * 7 + K + N)
    B, Tn, t0 = 5, 40, 3
    t1 = t0 + rows
    x = torch.randn(B * Tn, K, generator=g); W = torch.randn(K, N, generator=g) / 10
    Ww = make_weight(T(W))
    bias = T(torch.randn(N, generator=g))
    out = torch.full((B * Tn, N), 7.0, device=DEV)
    with paths() as log:
        ops.linear_rows(T(x), Ww, bias, out, B, Tn, t0, t1)
    assert log == [1], log
    ref = (bf(x).double() @ bf(W).double() + bias.double().cpu()).view(B, Tn, N)
    o3 = out.view(B, Tn, N)
    close(o3[:, t0:t1], ref[:, t0:t1], TOL, "rows fwd %dx%dx%d" % (rows, N, K))
    assert bool((o3[:, :t0] == 7.0).all()) and bool((o3[:, t1:] == 7.0).all())
    # input-gradient form (reduction over N), overwrite then accumulate
    if N % 8 == 0:
        dy = torch.randn(B * Tn, N, generator=g)
        prev = torch.randn(B * Tn, K, generator=g)
        refdx = (bf(dy).double() @ bf(W).double().T).view(B, Tn, K)
        for acc in (False, True):
            dx = T(prev).clone()
            with paths() as log:
                ops.linear_dx_rows(T(dy), Ww, dx, B, Tn, t0, t1, accumulate=acc)
            assert log == [1], log
            want = refdx + (prev.double().view(B, Tn, K) if acc else 0)
            close(dx.view(B, Tn, K)[:, t0:t1], want[:, t0:t1], TOL, "rows dX acc=%d" % acc)
            assert torch.equal(dx.view(B, Tn, K)[:, :t0].cpu(), prev.view(B, Tn, K)[:, :t0])
            assert torch.equal(dx.view(B, Tn, K)[:, t1:].cpu(), prev.view(B, Tn, K)[:, t1:])


@pytest.mark.parametrize("k,Cin,Cout,B,Tn", [(3, 128, 128, 2, 33), (5, 64, 96, 3, 17), (3, 2048, 128, 4, 160), (1, 32, 40, 2, 9),
                                             (10, 32, 64, 2, 70)])
def test_tile_conv1d_fwd_dx(k, Cin, Cout, B, Tn):
    from satt_amd import ops
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(k + Cin)
    x = torch.randn(B, Tn, Cin, generator=g); W = torch.randn(k, Cin, Cout, generator=g) / math.sqrt(k * Cin)
    dy = torch.randn(B, Tn, Cout, generator=g)
    xd, Wd, dyd = T(x).view(B * Tn, Cin), T(W), T(dy).view(B * Tn, Cout)
    Ww = make_weight(Wd)
    xr = bf(x).double().requires_grad_(True); Wr = bf(W).double()
    y = torch_ref.conv1d_same(xr, Wr)
    y.backward(bf(dy).double())
    out = torch.empty(B * Tn, Cout, device=DEV)
    with paths() as log:
        ops.conv1d(xd, Tn, Ww, out)
    assert log == [1], log
    close(out.view(B, Tn, Cout), y, TOL, "conv fwd")
    if Cout % 32 == 0:
        dx = torch.empty(B * Tn, Cin, device=DEV)
        with paths() as log:
            ops.conv1d_dx(dyd, Tn, Ww, dx)
        assert log == [1], log
        close(dx.view(B, Tn, Cin), xr.grad, TOL, "conv dx")


# conv_bank_fwd_k (path 3: 128 -> 128 channels, widths <= 16, samples of >= 36 steps): one partly filled row tile with two sample
# boundaries, the bench shape, sample lengths that put boundaries anywhere in a tile, fewer widths, a single long sample
BANK_CASES = [(16, 128, 128, 2, 40), (4, 32, 64, 3, 9), (5, 64, 32, 5, 37), (16, 128, 128, 32, 160), (16, 128, 128, 3, 177),
              (7, 128, 128, 5, 50), (16, 128, 128, 1, 300), (16, 128, 128, 7, 36), (16, 128, 128, 9, 35), (1, 128, 128, 2, 64)]


@pytest.mark.parametrize("ng,Cin,Cout,B,Tn", BANK_CASES)
def test_tile_conv_bank(ng, Cin, Cout, B, Tn):
    from satt_amd import ops
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(ng + Cin)
    x = torch.randn(B, Tn, Cin, generator=g)
    Ws = [torch.randn(k, Cin, Cout, generator=g) / math.sqrt(k * Cin) for k in range(1, ng + 1)]
    dy = torch.randn(B, Tn, ng * Cout, generator=g)
    flat = T(torch.cat([w.reshape(-1) for w in Ws]))
    st = torch.zeros(flat.numel(), dtype=torch.bfloat16, device=DEV); sn = torch.zeros_like(st)
    tab, off = [], 0
    for k in range(1, ng + 1):
        tab += [off, k, Cin, Cout]; off += k * Cin * Cout
    ops.shadow_pack(flat, torch.tensor(tab, dtype=torch.int64, device=DEV), ng, st, sn)
    Ww = ops.Weight(flat, st, sn)
    xd, dyd = T(x).view(B * Tn, Cin), T(dy).view(B * Tn, ng * Cout)
    xr = bf(x).double().requires_grad_(True)
    y = torch.cat([torch_ref.conv1d_same(xr, bf(w).double()) for w in Ws], dim=-1)
    y.backward(bf(dy).double())
    out = torch.empty(B * Tn, ng * Cout, device=DEV)
    out = torch.full((B * Tn, ng * Cout), float("nan"), device=DEV)
    with paths() as log:
        ops.conv_bank(xd, Tn, Ww, ng, out)
    assert log == [3 if (Cin, Cout) == (128, 128) and Tn >= 36 else 1], log
    close(out.view(B, Tn, -1), y, TOL, "bank fwd")
    if log == [3]:          # against the per-width launches on gemm_rk_k: the same bf16 operands, the same order of 32-wide K chunks
        per = torch.empty_like(out)
        off = 0
        for k in range(1, ng + 1):
            n = k * Cin * Cout
            wk = ops.Weight(flat[off:off + n].view(k, Cin, Cout), st[off:off + n], sn[off:off + n])
            ops.conv1d(xd, Tn, wk, per[:, (k - 1) * Cout:k * Cout])
            off += n
        assert float((out - per).abs().max()) <= 2e-6 * float(per.abs().max()), float((out - per).abs().max())
    dx = torch.zeros(B * Tn, Cin, device=DEV)
    with paths() as log:
        ops.conv_bank_dx(dyd, Tn, Ww, ng, dx)
    assert log == [3 if (Cin, Cout) == (128, 128) and Tn >= 36 else 1], log
    close(dx.view(B, Tn, Cin), xr.grad, TOL, "bank dx")
    if log == [3]:          # it ADDS into dx (one slab per width, summed in a fixed order): a second call doubles, bit for bit repeatable
        first = dx.clone()
        ops.conv_bank_dx(dyd, Tn, Ww, ng, dx)
        close(dx.view(B, Tn, Cin), 2 * xr.grad, TOL, "bank dx accumulates")
        again = torch.zeros_like(dx)
        ops.conv_bank_dx(dyd, Tn, Ww, ng, again)
        assert torch.equal(again, first)


@pytest.mark.parametrize("M,K,N", [(12800, 544, 1024), (5120, 128, 256), (700, 36, 72), (1000, 256, 164), (96, 64, 68),
                                   (5120, 2048, 128), (12800, 256, 256)])
def test_tile_linear_dw_with_bias_grad(M, K, N):
    from satt_amd import ops
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g); dy = torch.randn(M, N, generator=g)
    xd, dyd = T(x), T(dy)
    dW = torch.zeros(K, N, device=DEV); db = torch.zeros(N, device=DEV)
    with paths() as log:
        ops.linear_dw(xd, dyd, dW, db=db)
        ops.linear_dw(xd, dyd, dW, db=db)        # accumulates
    assert log == [2, 2], log
    close(dW, 2 * (bf(x).double().T @ bf(dy).double()), TOL, "linear_dw")
    close(db, 2 * dy.double().sum(0), 1e-5, "fused bias gradient")      # column sums are taken in fp32, unrounded


@pytest.mark.parametrize("shift", [-1, 1])
@pytest.mark.parametrize("B,Tn,Cx,N", [(32, 100, 256, 1024), (3, 9, 16, 72), (5, 37, 80, 72)])
def test_tile_shifted_dw(B, Tn, Cx, N, shift):
    """recurrent-weight gradient: x rows shifted by one step inside every sample, zero across sample boundaries"""
    from satt_amd import ops
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(B + Tn + Cx)
    x = torch.randn(B, Tn, Cx, generator=g); dy = torch.randn(B, Tn, N, generator=g)
    dW = torch.zeros(Cx, N, device=DEV); db = torch.zeros(N, device=DEV)
    with paths() as log:
        ops.shifted_dw(T(x).view(B * Tn, Cx), Tn, shift, T(dy).view(B * Tn, N), dW, db=db)
    assert log == [2], log
    xs = torch.zeros_like(x)
    if shift < 0:
        xs[:, 1:] = x[:, :-1]
    else:
        xs[:, :-1] = x[:, 1:]
    ref = bf(xs).double().reshape(B * Tn, Cx).T @ bf(dy).double().reshape(B * Tn, N)
    close(dW, ref, TOL, "shifted_dw")
    close(db, dy.double().reshape(B * Tn, N).sum(0), 1e-5, "bias gradient")


@pytest.mark.parametrize("k,Cin,Cout,B,Tn", [(3, 128, 128, 2, 33), (16, 128, 128, 4, 160), (5, 24, 72, 3, 17), (3, 2048, 128, 2, 40)])
def test_tile_conv1d_dw(k, Cin, Cout, B, Tn):
    from satt_amd import ops
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(k + Cin)
    x = torch.randn(B, Tn, Cin, generator=g); W = torch.randn(k, Cin, Cout, generator=g)
    dy = torch.randn(B, Tn, Cout, generator=g)
    xr = bf(x).double(); Wr = W.double().requires_grad_(True)
    torch_ref.conv1d_same(xr, Wr).backward(bf(dy).double())
    dW = torch.zeros(k, Cin, Cout, device=DEV)
    with paths() as log:
        ops.conv1d_dw(T(x).view(B * Tn, Cin), Tn, T(dy).view(B * Tn, Cout), dW)
    assert log == [2], log
    close(dW, Wr.grad, TOL, "conv dw")


@pytest.mark.parametrize("ng,Cin,Cout,B,Tn", [(16, 128, 128, 32, 160), (4, 32, 72, 3, 9), (5, 40, 96, 5, 37)])
def test_tile_conv_bank_dw_one_launch(ng, Cin, Cout, B, Tn):
    """weight gradients of every width of the conv bank in ONE launch (tiles enumerated group by group)"""
    from satt_amd import ops
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(ng + Cin)
    x = torch.randn(B, Tn, Cin, generator=g); dy = torch.randn(B, Tn, ng * Cout, generator=g)
    Ws = [torch.zeros(k, Cin, Cout, dtype=torch.float64, requires_grad=True) for k in range(1, ng + 1)]
    y = torch.cat([torch_ref.conv1d_same(bf(x).double(), w) for w in Ws], dim=-1)
    y.backward(bf(dy).double())
    ref = torch.cat([w.grad.reshape(-1) for w in Ws])
    dW = torch.zeros(ref.numel(), device=DEV)
    with paths() as log:
        ops.conv_bank_dw(T(x).view(B * Tn, Cin), Tn, T(dy).view(B * Tn, ng * Cout), dW, ng)
    assert log == [2], log
    close(dW, ref, TOL, "bank dW")


def test_split_reductions_use_slabs_and_overwrite():
    """split-K forward conv (the 2048-channel projection) through the slab workspace: C is REPLACED without being zeroed
    first, and the result is bit-identical from run to run (no atomics)"""
    from satt_amd import ops
    ops.set_precision("bf16")
    g = torch.Generator().manual_seed(5)
    B, Tn, Cin, Cout, k = 4, 160, 2048, 128, 3
    x = torch.randn(B, Tn, Cin, generator=g); W = torch.randn(k, Cin, Cout, generator=g) / math.sqrt(k * Cin)
    Ww = make_weight(T(W))
    xd = T(x).view(B * Tn, Cin)
    outs = []
    for _ in range(2):
        out = torch.full((B * Tn, Cout), 7.0, device=DEV)          # stale contents must not leak into the result
        with paths() as log:
            ops.conv1d(xd, Tn, Ww, out)
        assert log == [1], log
        outs.append(out.clone())
    close(outs[0].view(B, Tn, Cout), torch_ref.conv1d_same(bf(x).double(), bf(W).double()), TOL, "split-K conv (slabs)")
    assert torch.equal(outs[0], outs[1])


def test_generic_fallbacks_still_serve_what_the_tiles_decline():
    """no shadow, odd sizes, fp32 mode: the generic kernel (path 0) with a separate bias column sum"""
    from satt_amd import ops
    g = torch.Generator().manual_seed(0)
    M, K, N = 90, 41, 23
    x = torch.randn(M, K, generator=g); W = torch.randn(K, N, generator=g); dy = torch.randn(M, N, generator=g)
    for prec, tol in (("bf16", 2e-3), ("f32", 1e-5)):
        ops.set_precision(prec)
        r = bf if prec == "bf16" else (lambda t: t)
        out = torch.empty(M, N, device=DEV)
        dW = torch.zeros(K, N, device=DEV); db = torch.zeros(N, device=DEV)
        with paths() as log:
            ops.linear(T(x), make_weight(T(W)), None, out)
            ops.linear_dw(T(x), T(dy), dW, db=db)
        assert log == [0, 0], log
        close(out, r(x).double() @ r(W).double(), tol, prec + " linear (generic)")
        close(dW, r(x).double().T @ r(dy).double(), tol, prec + " linear_dw (generic)")
        close(db, dy.double().sum(0), 1e-5, prec + " bias gradient (separate launch)")
    ops.set_precision("bf16")
