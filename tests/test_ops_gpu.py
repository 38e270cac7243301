"""Op-level GPU parity tests: each HIP entry point of include/satt_hip.h against a plain PyTorch fp32/fp64
reference of the same op on seeded inputs (tolerances stated per test)."""
import math

import numpy as np
import pytest
import torch

import satt_amd  # noqa: F401
from oracle import rng, torch_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32, device=DEV).contiguous()


def close(a, b, tol, what=""):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
    print("%-24s rel_err=%.3e" % (what, err))
    assert err < tol, (what, err)


def test_arch_and_version():
    from satt_amd import _lib
    l = _lib.lib()
    assert l.satt_version() >= 100
    assert l.satt_arch_supported(0) == 1


@pytest.mark.parametrize("prec,tol", [("f32", 2e-6), ("bf16", 2e-2)])
@pytest.mark.parametrize("M,N,K", [(70, 50, 33), (129, 64, 200), (5, 161, 64), (300, 130, 1000),
                                   # 16-byte-vectorisable shapes: running-pointer fast path, K tails of the 32- and
                                   # 64-deep tiles, large grids (shallow tile) and one-workgroup-per-CU grids (deep tile)
                                   (256, 128, 544), (64, 256, 96), (1000, 512, 36), (12, 1024, 260), (2048, 1024, 132)])
def test_linear_fwd_bwd(prec, tol, M, N, K):
    from satt_amd import ops
    ops.set_precision(prec)
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g); W = torch.randn(K, N, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    xd, Wd, bd, rd = T(x), T(W), T(b), T(res)
    if prec == "bf16":   # compare against the same bf16-rounded operands
        x = x.bfloat16().float(); W = W.bfloat16().float()
    out = torch.empty(M, N, device=DEV)
    ops.linear(xd, Wd, bd, out, act=ops.ACT_TANH, residual=rd)
    close(out, torch.tanh(x.double() @ W.double() + b.double()) + res.double(), tol if prec == "f32" else 1e-3, "linear tanh+res")
    dy = torch.randn(M, N, generator=g)
    dyd = T(dy)
    if prec == "bf16":
        dy = dy.bfloat16().float()
    dx = torch.empty(M, K, device=DEV)
    ops.linear_dx(dyd, Wd, dx)
    close(dx, dy.double() @ W.double().T, tol if prec == "f32" else 1e-3, "linear_dx")
    dW = torch.zeros(K, N, device=DEV)
    ops.linear_dw(xd, dyd, dW)
    ops.linear_dw(xd, dyd, dW)      # accumulates
    close(dW, 2 * (x.double().T @ dy.double()), 1e-5 if prec == "f32" else 1e-3, "linear_dw")


def test_gemm_strided_views_and_dropout():
    from satt_amd import ops
    ops.set_precision("f32")
    g = torch.Generator().manual_seed(0)
    M, K, N = 90, 40, 24
    x = torch.randn(M, K + 8, generator=g); W = torch.randn(K + 3, N + 5, generator=g)
    xd, Wd = T(x), T(W)
    big = torch.zeros(M, 3 * N, device=DEV)
    seed = torch.tensor([123], dtype=torch.int32, device=DEV)
    ops.linear(xd[:, 8:], Wd[3:, 5:], None, big[:, N:2 * N], act=ops.ACT_RELU, drop=ops.Drop(0.5, 9, seed))
    ref = torch.relu(x[:, 8:].double() @ W[3:, 5:].double())
    mask = torch.from_numpy(rng.keep_mask(123, 9, (M, N), 0.5))
    close(big[:, N:2 * N], ref * mask * 2.0, 2e-6, "strided+dropout")
    assert float(big[:, :N].abs().max()) == 0 and float(big[:, 2 * N:].abs().max()) == 0


@pytest.mark.parametrize("prec,tol", [("f32", 3e-6), ("bf16", 2e-3)])
@pytest.mark.parametrize("k,Cin,Cout,B,Tn", [(1, 8, 8, 2, 5), (4, 24, 40, 3, 17), (10, 16, 24, 2, 70), (3, 136, 72, 2, 33)])
def test_conv1d_fwd_bwd(prec, tol, k, Cin, Cout, B, Tn):
    from satt_amd import ops
    ops.set_precision(prec)
    g = torch.Generator().manual_seed(k + Cin)
    x = torch.randn(B, Tn, Cin, generator=g); W = torch.randn(k, Cin, Cout, generator=g) / math.sqrt(k * Cin)
    dy = torch.randn(B, Tn, Cout, generator=g)
    xd, Wd, dyd = T(x).view(B * Tn, Cin), T(W), T(dy).view(B * Tn, Cout)
    if prec == "bf16":
        x = x.bfloat16().float(); W = W.bfloat16().float(); dy = dy.bfloat16().float()
    xr = x.double().requires_grad_(True); Wr = W.double().requires_grad_(True)
    y = torch_ref.conv1d_same(xr, Wr)
    y.backward(dy.double())
    out = torch.empty(B * Tn, Cout, device=DEV)
    ops.conv1d(xd, Tn, Wd, out)
    close(out.view(B, Tn, Cout), y, tol, "conv fwd")
    dx = torch.empty(B * Tn, Cin, device=DEV)
    ops.conv1d_dx(dyd, Tn, Wd, dx)
    close(dx.view(B, Tn, Cin), xr.grad, tol, "conv dx")
    dW = torch.zeros(k, Cin, Cout, device=DEV)
    ops.conv1d_dw(xd, Tn, dyd, dW)
    close(dW, Wr.grad, 1e-5 if prec == "f32" else tol, "conv dw")


@pytest.mark.parametrize("prec,tol", [("f32", 3e-6), ("bf16", 2e-3)])
@pytest.mark.parametrize("ng,Cin,Cout,B,Tn", [(4, 8, 8, 3, 9), (16, 128, 128, 2, 40), (5, 40, 24, 5, 37)])
def test_conv_bank_one_launch(prec, tol, ng, Cin, Cout, B, Tn):
    """the grouped conv-bank launch (all widths 1..ng at once) against the fp64 oracle convolution per width"""
    from satt_amd import ops
    ops.set_precision(prec)
    g = torch.Generator().manual_seed(ng + Cin)
    x = torch.randn(B, Tn, Cin, generator=g)
    Ws = [torch.randn(k, Cin, Cout, generator=g) / math.sqrt(k * Cin) for k in range(1, ng + 1)]
    dy = torch.randn(B, Tn, ng * Cout, generator=g)
    if prec == "bf16":
        x = x.bfloat16().float(); Ws = [w.bfloat16().float() for w in Ws]; dy = dy.bfloat16().float()
    Wall = T(torch.cat([w.reshape(-1) for w in Ws]))
    xd, dyd = T(x).view(B * Tn, Cin), T(dy).view(B * Tn, ng * Cout)
    xr = x.double().requires_grad_(True)
    y = torch.cat([torch_ref.conv1d_same(xr, w.double()) for w in Ws], dim=-1)
    y.backward(dy.double())
    out = torch.empty(B * Tn, ng * Cout, device=DEV)
    ops.conv_bank(xd, Tn, Wall, ng, out)
    close(out.view(B, Tn, ng * Cout), y, tol, "conv bank fwd")
    dx = torch.zeros(B * Tn, Cin, device=DEV)
    ops.conv_bank_dx(dyd, Tn, Wall, ng, dx)
    close(dx.view(B, Tn, Cin), xr.grad, tol * 2, "conv bank dx")


def test_dropout_and_softmax_rows():
    from satt_amd import ops
    from oracle import rng
    g = torch.Generator().manual_seed(3)
    rows, cols = 37, 50
    x = torch.randn(rows, cols, generator=g)
    seed = torch.tensor([1234], dtype=torch.int32, device=DEV)
    d = ops.Drop(0.5, 17, seed)
    y = torch.empty(rows, cols, device=DEV)
    ops.dropout(T(x), y, d)
    keep = torch.from_numpy(rng.keep_mask(1234, 17, (rows, cols), 0.5)).float()
    close(y, x * keep * 2.0, 1e-6, "dropout")
    # strided row softmax (KV-cache query row): columns beyond `n` untouched
    s = torch.randn(6, 40, generator=g)
    p = torch.full((6, 40), -1.0, device=DEV)
    n = 23
    ops.softmax_rows(T(s), p, 6, n, 0.37)
    close(p[:, :n], torch.softmax(s[:, :n].double() * 0.37, dim=-1), 1e-6, "softmax rows")
    assert float((p[:, n:] + 1.0).abs().max()) == 0.0


def test_shifted_dw():
    from satt_amd import ops
    ops.set_precision("f32")
    g = torch.Generator().manual_seed(5)
    B, Tn, Cx, N = 3, 11, 20, 28
    x = torch.randn(B, Tn, Cx, generator=g); dy = torch.randn(B, Tn, N, generator=g)
    for shift in (-1, 1):
        dW = torch.zeros(Cx, N, device=DEV)
        ops.shifted_dw(T(x).view(-1, Cx), Tn, shift, T(dy).view(-1, N), dW)
        xs = torch.zeros_like(x)
        if shift == -1:
            xs[:, 1:] = x[:, :-1]
        else:
            xs[:, :-1] = x[:, 1:]
        close(dW, torch.einsum("btc,btn->cn", xs.double(), dy.double()), 1e-5, "shifted_dw %d" % shift)


@pytest.mark.parametrize("act", [0, 1])
def test_batchnorm_fwd_bwd(act):
    from satt_amd import ops
    g = torch.Generator().manual_seed(3)
    rows, Cc = 300, 70
    x = torch.randn(rows, Cc, generator=g) * 2 + 0.5
    gamma = torch.rand(Cc, generator=g) + 0.5; beta = torch.randn(Cc, generator=g) * 0.3
    dy = torch.randn(rows, Cc, generator=g)
    xr = x.double().requires_grad_(True); gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    y = torch_ref.batch_norm(xr[None], gr, br, 1e-3, True)[0]
    if act:
        y = torch.relu(y)
    y.backward(dy.double())
    xd = T(x); yd = torch.empty_like(xd)
    mean, rstd = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
    mm, mv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    ws = ops.bn_ws(rows, Cc, DEV)
    ops.bn_fwd(xd, T(gamma), T(beta), yd, mean, rstd, mm, mv, ws, 1e-3, 0.99, act)
    close(yd, y, 5e-6, "bn fwd")
    close(mm, 0.01 * x.double().mean(0), 1e-5, "bn moving mean")
    dx = torch.empty_like(xd); dg, db = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    ops.bn_bwd(T(dy), xd, T(gamma), T(beta), mean, rstd, dx, dg, db, ws, act)
    close(dx, xr.grad, 2e-5, "bn dx"); close(dg, gr.grad, 2e-5, "bn dgamma"); close(db, br.grad, 2e-5, "bn dbeta")


def test_bn_relu_maxpool_fused():
    """BatchNorm (training statistics) + ReLU + MaxPooling1D(2, 1, SAME) of the conv bank as one forward pass and a fused
    backward pair that recomputes the activated values (csrc/elementwise.hip) against autograd through the same three ops"""
    from satt_amd import ops
    g = torch.Generator().manual_seed(8)
    B, Tn, Cc = 3, 37, 72
    rows = B * Tn
    x = torch.randn(rows, Cc, generator=g) * 2 + 0.3
    x[5] = x[4]; x[40:43] = x[39]            # exact ties between neighbouring steps
    gamma = torch.randn(Cc, generator=g); beta = torch.randn(Cc, generator=g) * 0.3      # negative gammas as well
    dmp = torch.randn(rows, Cc, generator=g)
    xr = x.double().requires_grad_(True); gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    y = torch.relu(torch_ref.batch_norm(xr.view(B, Tn, Cc), gr, br, 1e-3, True))
    mp = torch.maximum(y, torch.cat([y[:, 1:], y[:, -1:]], 1))
    xd = T(x); mpd = torch.empty_like(xd)
    mean, rstd = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
    mm, mv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    ws = ops.bn_ws(rows, Cc, DEV)
    assert ops.bn_maxpool_fwd(xd, T(gamma), T(beta), mpd, mean, rstd, mm, mv, ws, B, Tn, 1e-3, 0.99, ops.ACT_RELU)
    close(mpd, mp.reshape(rows, Cc), 5e-6, "bn + relu + maxpool fwd")
    # reference backward with the library's own tie rule (first element of the window): separate kernels on the stored activations
    yd = torch.empty_like(xd); m2, r2 = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV); ws2 = ops.bn_ws(rows, Cc, DEV)
    ops.bn_fwd(xd, T(gamma), T(beta), yd, m2, r2, torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV), ws2, 1e-3, 0.99, ops.ACT_RELU)
    dyd = torch.empty_like(xd); ops.maxpool_bwd(T(dmp), yd, dyd, B, Tn, Cc)
    dx_ref = torch.empty_like(xd); dg_ref, db_ref = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    ops.bn_bwd(dyd, xd, T(gamma), T(beta), m2, r2, dx_ref, dg_ref, db_ref, ws2, ops.ACT_RELU)
    dx = torch.empty_like(xd); dg, db = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    ops.maxpool_bn_bwd(T(dmp), xd, T(gamma), T(beta), mean, rstd, dx, dg, db, ws, torch.empty_like(xd), B, Tn, ops.ACT_RELU)
    close(dx, dx_ref.double().cpu(), 2e-6, "fused dx"); close(dg, dg_ref.double().cpu(), 2e-6, "fused dgamma"); close(db, db_ref.double().cpu(), 2e-6, "fused dbeta")
    # and against autograd where no tie is involved: the sums over everything
    mp.backward(dmp.double().view(B, Tn, Cc))
    close(dg, gr.grad, 5e-4, "fused dgamma vs autograd"); close(db, br.grad, 5e-4, "fused dbeta vs autograd")


@pytest.mark.parametrize("rows,Cc,act", [(5120, 128, "relu"), (333, 72, "none"), (64, 8, "tanh"), (1, 200, "relu"), (4097, 130, "none")])
def test_bn_one_launch_pair(rows, Cc, act):
    """satt_bn_fwd_fused / satt_bn_bwd_fused (one launch each, last-arriver merge + in-kernel release) against float64
    autograd BatchNorm, launched repeatedly on ONE state (the barrier words must return to zero), strided views included"""
    from satt_amd import ops
    g = torch.Generator().manual_seed(rows + Cc)
    A = {"relu": ops.ACT_RELU, "none": ops.ACT_NONE, "tanh": ops.ACT_TANH}[act]
    f = {"relu": torch.relu, "none": (lambda t: t), "tanh": torch.tanh}[act]
    x = torch.randn(rows, Cc, generator=g) * 1.7 + 3.0
    gamma = torch.randn(Cc, generator=g); beta = torch.randn(Cc, generator=g) * 0.3
    dy = torch.randn(rows, Cc, generator=g)
    xr = x.double().requires_grad_(True); gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    mu = xr.mean(0); var = ((xr - mu) ** 2).mean(0)
    y = f((xr - mu) / torch.sqrt(var + 1e-3) * gr + br)
    y.backward(dy.double())
    xbuf = torch.zeros(rows, Cc + 5, device=DEV); xbuf[:, :Cc] = T(x)              # leading dimension > C
    xd = xbuf[:, :Cc]
    st = ops.bn_fused_state(rows, Cc, DEV)
    st[0].fill_(float("nan"))
    mm, mv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    for rep in range(3):
        yd = torch.full((rows, Cc), float("nan"), device=DEV)
        mean, rstd = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
        assert ops.bn_fwd_fused(xd, T(gamma), T(beta), yd, mean, rstd, mm, mv, st, 1e-3, 0.99, A)
        close(yd, y, 2e-5, "one-launch bn fwd"); close(mean, mu, 1e-5, "batch mean")
        dx = torch.full((rows, Cc), float("nan"), device=DEV); dg, db = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
        assert ops.bn_bwd_fused(T(dy), xd, T(gamma), T(beta), mean, rstd, dx, dg, db, st, A)
        if rows > 1:
            close(dx, xr.grad, 2e-4, "one-launch bn dx")
        close(dg, gr.grad, 2e-4, "one-launch bn dgamma"); close(db, br.grad, 2e-4, "one-launch bn dbeta")
        assert int(st[1].abs().sum()) == 0                                              # barrier words back to zero
    close(mm, (1 - 0.99 ** 3) * mu.detach(), 1e-4, "moving mean after three updates")
    # the three-launch path gives the same numbers
    y3 = torch.empty(rows, Cc, device=DEV); m3, r3 = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
    ops.bn_fwd(xd, T(gamma), T(beta), y3, m3, r3, torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV), ops.bn_ws(rows, Cc, DEV), 1e-3, 0.99, A)
    close(yd, y3.double().cpu(), 2e-6, "one launch vs three launches")


def test_bn_one_launch_declines_large_grids():
    from satt_amd import ops
    rows, Cc = 25600, 512          # the post-net's shape: 1600 workgroups would have to wait on each other
    st = ops.bn_fused_state(rows, Cc, DEV)
    x = torch.randn(rows, Cc, device=DEV); y = torch.empty_like(x)
    v = lambda: torch.ones(Cc, device=DEV)
    assert ops.bn_fwd_fused(x, v(), v(), y, v(), v(), v(), v(), st, 1e-3, 0.99, ops.ACT_NONE) is False


def test_maxpool_highway_misc():
    from satt_amd import ops
    g = torch.Generator().manual_seed(4)
    B, Tn, Cc = 3, 9, 20
    x = torch.randn(B, Tn, Cc, generator=g); dy = torch.randn(B, Tn, Cc, generator=g)
    xr = x.double().requires_grad_(True)
    y = torch.maximum(xr, torch.cat([xr[:, 1:], xr[:, -1:]], 1)); y.backward(dy.double())
    yd = torch.empty(B, Tn, Cc, device=DEV); ops.maxpool_fwd(T(x), yd, B, Tn, Cc)
    close(yd, y, 1e-7, "maxpool fwd")
    dx = torch.empty(B, Tn, Cc, device=DEV); ops.maxpool_bwd(T(dy), T(x), dx, B, Tn, Cc)
    close(dx, xr.grad, 1e-6, "maxpool bwd")
    for Cs in (21, 64):      # scalar form (C % 4 != 0) and a second float4 shape, with ties (repeated values)
        xs = torch.randint(0, 3, (2, 7, Cs), generator=g).float(); dys = torch.randn(2, 7, Cs, generator=g)
        # ties go to the FIRST element of the window (dx[t] takes dy[t] when x[t] >= x[t+1], dy[t-1] only when x[t] > x[t-1])
        ref = torch.zeros(2, 7, Cs, dtype=torch.float64)
        for t in range(7):
            if t + 1 < 7:
                first = xs[:, t] >= xs[:, t + 1]
                ref[:, t] += torch.where(first, dys[:, t].double(), torch.zeros(()).double())
                ref[:, t + 1] += torch.where(first, torch.zeros(()).double(), dys[:, t].double())
            else:
                ref[:, t] += dys[:, t].double()
        dxs = torch.empty(2, 7, Cs, device=DEV); ops.maxpool_bwd(T(dys), T(xs), dxs, 2, 7, Cs)
        close(dxs, ref, 1e-6, "maxpool bwd ties C=%d" % Cs)
    rows, H = 50, 24
    z = torch.randn(rows, 2 * H, generator=g); xx = torch.randn(rows, H, generator=g); dyy = torch.randn(rows, H, generator=g)
    zr = z.double().requires_grad_(True); xr = xx.double().requires_grad_(True)
    tt = torch.sigmoid(zr[:, H:]); y = torch.relu(zr[:, :H]) * tt + xr * (1 - tt); y.backward(dyy.double())
    yd = torch.empty(rows, H, device=DEV); ops.highway_fwd(T(z), T(xx), yd); close(yd, y, 2e-6, "highway fwd")
    dz, dx = torch.empty(rows, 2 * H, device=DEV), torch.empty(rows, H, device=DEV)
    ops.highway_bwd(T(dyy), T(z), T(xx), dz, dx)
    close(dz, zr.grad, 5e-6, "highway dz"); close(dx, xr.grad, 5e-6, "highway dx")
    out = torch.ones(2 * H, device=DEV); ops.colsum(T(z), out, accumulate=True)
    close(out, 1 + z.double().sum(0), 1e-5, "colsum")
    ids = torch.randint(3, 13, (4, 6), generator=g)
    table = torch.randn(10, 8, generator=g); o = torch.empty(24, 8, device=DEV)
    ops.embedding_fwd(ids.to(DEV), T(table), o, offset=3); close(o, table[ids.view(-1) - 3], 1e-7, "embedding fwd")
    dt = torch.zeros(10, 8, device=DEV); do = torch.randn(24, 8, generator=g)
    ops.embedding_bwd(ids.to(DEV), T(do), dt, offset=3)
    ref = torch.zeros(10, 8, dtype=torch.float64); ref.index_add_(0, ids.view(-1) - 3, do.double())
    close(dt, ref, 1e-5, "embedding bwd")
    dta = torch.zeros(10, 8, device=DEV); ops.embedding_bwd(ids.to(DEV), T(do), dta, offset=3, atomic=False)
    close(dta, ref, 1e-5, "embedding bwd (deterministic form)")
    # the benchmark's shape (5120 tokens, 256 symbols x 256): accumulates onto the existing table gradient, bit-identical
    # from run to run (one workgroup per table row adds its tokens in ascending order)
    ids2 = torch.randint(0, 256, (5120,), generator=g); ids2[:700] = 7; do2 = torch.randn(5120, 256, generator=g)
    base = torch.randn(256, 256, generator=g)
    ref2 = base.double().clone(); ref2.index_add_(0, ids2, do2.double())
    d1 = T(base).clone(); ops.embedding_bwd(ids2.to(DEV), T(do2), d1, atomic=False)
    d2 = T(base).clone(); ops.embedding_bwd(ids2.to(DEV), T(do2), d2, atomic=False)
    close(d1, ref2, 2e-6, "embedding bwd rows"); assert torch.equal(d1, d2)
    # atomic form: runs of equal ids inside a group of 8 consecutive tokens are merged before the table is touched (the padding
    # tails of a batch); n not a multiple of 8, runs that straddle groups, a run at the very end
    ids3 = ids2[:5117].clone(); ids3[3000:3400] = 0; ids3[5100:] = 0
    ref3 = base.double().clone(); ref3.index_add_(0, ids3, do2[:5117].double())
    d3 = T(base).clone(); ops.embedding_bwd(ids3.to(DEV), T(do2[:5117].contiguous()), d3)
    close(d3, ref3, 2e-5, "embedding bwd (atomic form, merged runs)")
    # activation backward, plain and with the activation output given as (z - res) (transformer tail x + tanh(Dense(.)))
    u = torch.randn(37, 24, generator=g); res = torch.randn(37, 24, generator=g) * 3; dyz = torch.randn(37, 24, generator=g)
    th = torch.tanh(u)
    d1 = torch.empty(37, 24, device=DEV); ops.act_bwd(T(dyz), T(th), d1, ops.ACT_TANH)
    close(d1, dyz.double() * (1 - torch.tanh(u.double()) ** 2), 2e-6, "act_bwd tanh")
    d2 = torch.empty(37, 24, device=DEV); ops.act_bwd_res(T(dyz), T(th + res), T(res), d2, ops.ACT_TANH)
    close(d2, dyz.double() * (1 - torch.tanh(u.double()) ** 2), 2e-6, "act_bwd_res tanh")
    w = torch.randn(33, 20, generator=g)
    wb = torch.empty(33, 20, dtype=torch.bfloat16, device=DEV); ops.to_bf16(T(w), wb)
    assert torch.equal(wb.cpu(), w.bfloat16())
    wbt = torch.empty(20, 33, dtype=torch.bfloat16, device=DEV); ops.to_bf16(T(w), wbt, transpose=True)
    assert torch.equal(wbt.cpu(), w.bfloat16().T)


@pytest.mark.parametrize("causal,rate,Tn", [(False, 0.0, 9), (True, 0.05, 70), (False, 0.05, 160), (True, 0.05, 400)])
def test_softmax_fwd_bwd(causal, rate, Tn):
    from satt_amd import ops
    g = torch.Generator().manual_seed(Tn)
    nbh = 6
    s = torch.randn(nbh, Tn, Tn, generator=g) * 3; dpd = torch.randn(nbh, Tn, Tn, generator=g)
    scale = 0.25
    sr = s.double().requires_grad_(True)
    x = sr * scale
    if causal:
        x = torch.where(torch.tril(torch.ones(Tn, Tn, dtype=torch.bool)), x, torch.full_like(x, float("-inf")))
    p = torch.softmax(x, -1)
    mask = torch.from_numpy(rng.keep_mask(77, 16, (nbh, Tn, Tn), rate)).double()
    pdrop = p * mask / (1 - rate)
    pdrop.backward(dpd.double())
    seed = torch.tensor([77], dtype=torch.int32, device=DEV)
    drop = ops.Drop(rate, 16, seed)
    pd_, pdd = torch.empty(nbh, Tn, Tn, device=DEV), torch.empty(nbh, Tn, Tn, device=DEV)
    ops.softmax_fwd(T(s), pd_, pdd, nbh, Tn, scale, causal, drop)
    close(pd_, p, 3e-6, "softmax p"); close(pdd, pdrop, 3e-6, "softmax pd")
    ds = torch.empty(nbh, Tn, Tn, device=DEV)
    ops.softmax_bwd(T(dpd), pd_, ds, nbh, Tn, scale, causal, drop)
    close(ds, sr.grad, 2e-5, "softmax ds")


@pytest.mark.parametrize("H,B,Tn,ndir,use_len", [(8, 3, 7, 2, True), (40, 4, 19, 2, True), (64, 3, 12, 1, False),
                                                 (128, 2, 9, 2, True), (160, 2, 6, 2, True)])
@pytest.mark.parametrize("training", [True, False])
def test_lstm_fwd_bwd(H, B, Tn, ndir, use_len, training):
    from satt_amd import ops
    from common import bf16_round
    g = np.random.default_rng(H + Tn)
    xg = g.normal(0, 1, (ndir, B, Tn, 4 * H)).astype(np.float32)
    Wh = bf16_round(g.normal(0, 1.0 / math.sqrt(H), (ndir, H, 4 * H)))
    lens = g.integers(2, Tn + 1, B) if use_len else None
    if use_len:
        lens[0] = Tn
    dh = g.normal(0, 1, (B, Tn, ndir * H)).astype(np.float32)
    zc, zh, seed = 0.1, 0.15, 31
    sc, sh = (3, 5)[:ndir], (4, 6)[:ndir]
    # oracle: zero input weights, xg added through the bias path
    xr = torch.tensor(xg, dtype=torch.float64, requires_grad=True)
    outs = []
    for d in range(ndir):
        W = torch.cat([torch.eye(4 * H, dtype=torch.float64), torch.tensor(Wh[d], dtype=torch.float64)], 0)
        outs.append(torch_ref.zoneout_lstm_seq(xr[d], W, torch.zeros(4 * H, dtype=torch.float64), H,
                                               torch.tensor(lens) if use_len else None, d == 1, zc, zh, training, seed,
                                               (sc[d], sh[d])))
    y = torch.cat(outs, -1)
    y.backward(torch.tensor(dh, dtype=torch.float64))
    Whb = torch.tensor(Wh).to(torch.bfloat16).to(DEV).contiguous()
    WhT = torch.tensor(Wh).transpose(1, 2).contiguous().to(torch.bfloat16).to(DEV)
    hout = torch.full((B * Tn, ndir * H), 7.0, device=DEV)
    e = lambda *s: torch.full(s, 9.0, device=DEV)
    gates, cn, cs, hs = e(ndir, B * Tn, 4 * H), e(ndir, B * Tn, H), e(ndir, B * Tn, H), e(ndir, B * Tn, H)
    seedt = torch.tensor([seed], dtype=torch.int32, device=DEV)
    lent = torch.tensor(lens, dtype=torch.int64, device=DEV) if use_len else None
    ops.lstm_fwd(T(xg), Whb, lent, ndir, B, Tn, H, training, zc, zh, seedt, sc, sh, hout, gates, cn, cs, hs)
    close(hout.view(B, Tn, -1), y, 1e-5, "lstm fwd")
    dxg = e(ndir, B * Tn, 4 * H)
    ops.lstm_bwd(T(dh).view(B * Tn, -1), WhT, lent, ndir, B, Tn, H, training, zc, zh, seedt, sc, sh, gates, cn, cs, dxg)
    close(dxg.view(ndir, B, Tn, 4 * H), xr.grad, 5e-5, "lstm dxg")


@pytest.mark.parametrize("B,Td,r,nm", [(3, 7, 2, 5), (2, 37, 2, 80), (2, 5, 3, 70)])
def test_loss_kernels(B, Td, r, nm):
    from satt_amd import ops
    g = np.random.default_rng(9)
    Tm, NO = Td * r, r * nm + 1
    y = g.normal(0, 1, (B * Td, NO)).astype(np.float32)
    tgt = g.normal(0, 1, (B, Tm, nm)).astype(np.float32)
    sm = (g.random((B, Tm)) > 0.3).astype(np.float32); bm = (g.random((B, Td)) > 0.3).astype(np.float32)
    done = (g.random((B, Td)) > 0.7).astype(np.float32)
    yr = torch.tensor(y, dtype=torch.float64, requires_grad=True)
    mel = yr[:, :-1].reshape(B, Tm, nm); stop = yr[:, -1:].reshape(B, Td, 1)
    bt = dict(mel=torch.tensor(tgt, dtype=torch.float64), spec_loss_mask=torch.tensor(sm, dtype=torch.float64),
              done=torch.tensor(done, dtype=torch.float64), binary_loss_mask=torch.tensor(bm, dtype=torch.float64))
    ml, dl = torch_ref.losses(mel, stop, bt)
    (ml + dl).backward()
    yd = T(y); dy = torch.empty_like(yd); losses = torch.zeros(3, device=DEV); ws = torch.zeros(4, device=DEV)
    ops.loss_fwd_bwd(yd, NO, T(tgt), T(sm), yd[:, NO - 1:], NO, T(done), T(bm), B, Tm, nm, Td, False, losses, dy, NO,
                     dy[:, NO - 1:], NO, ws)
    close(losses, torch.stack([ml, dl, ml + dl]), 2e-6, "losses"); close(dy, yr.grad, 2e-6, "loss grad")
    # the split form of the training step: mask sums first (any time before), then ONE launch; twice on the same workspace
    dy2 = torch.empty_like(yd); losses2 = torch.zeros(3, device=DEV); ws8 = torch.full((8,), 7.0, device=DEV)
    for _ in range(2):
        ops.loss_mask_sums(T(sm), T(bm), B, Tm, Td, ws8)
        ops.loss_fwd_bwd_presummed(yd, NO, T(tgt), T(sm), yd[:, NO - 1:], NO, T(done), T(bm), B, Tm, nm, Td, False, losses2, dy2, NO,
                                   dy2[:, NO - 1:], NO, ws8)
    close(losses2, torch.stack([ml, dl, ml + dl]), 2e-6, "losses (presummed)"); close(dy2, yr.grad, 2e-6, "loss grad (presummed)")
    # padded rows [mel | stop | pad] (the engine's layout: 16-byte rows): same values, pad columns of the gradient zero-filled
    NOp = NO + (-NO) % 8
    yp = torch.full((B * Td, NOp), float("nan"), device=DEV); yp[:, :NO] = yd
    dyp = torch.full((B * Td, NOp), float("nan"), device=DEV); losses3 = torch.zeros(3, device=DEV)
    ops.loss_mask_sums(T(sm), T(bm), B, Tm, Td, ws8)
    ops.loss_fwd_bwd_presummed(yp[:, :NO], NOp, T(tgt), T(sm), yp[:, NO - 1:], NOp, T(done), T(bm), B, Tm, nm, Td, False, losses3,
                               dyp[:, :NO], NOp, dyp[:, NO - 1:], NOp, ws8)
    close(losses3, torch.stack([ml, dl, ml + dl]), 2e-6, "losses (padded rows)")
    close(dyp[:, :NO], yr.grad, 2e-6, "loss grad (padded rows)")
    assert NOp > NO and bool((dyp[:, NO:] == 0).all())


def test_loss_and_adam():
    from satt_amd import ops
    g = np.random.default_rng(9)
    # optimiser: 3 steps against the oracle's TF-Adam
    n = 1000
    p0 = g.normal(0, 1, n).astype(np.float32)
    P = {"w": torch.tensor(p0, dtype=torch.float64)}; m = {"w": torch.zeros(n, dtype=torch.float64)}; v = {"w": torch.zeros(n, dtype=torch.float64)}
    pd_, md, vd = T(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    state = ops.opt_state(DEV); step = torch.zeros(1, dtype=torch.int32, device=DEV)
    seedt = torch.zeros(1, dtype=torch.int32, device=DEV)
    for t in range(1, 4):
        gr = g.normal(0, 0.2 * t, n).astype(np.float32)
        lr = torch_ref.learning_rate(5e-4, t - 1)
        gn = torch_ref.clip_and_adam(P, {"w": torch.tensor(gr, dtype=torch.float64)}, m, v, t, lr)
        gd = T(gr)
        ops.sumsq(gd, state)
        ops.adam_step(pd_, gd, md, vd, state, step, seedt, 5e-4, True, 1.0, 0.9, 0.999, 1e-8, 1.0, 1.0)
        torch.cuda.synchronize()
        assert abs(float(state[1]) - gn) / gn < 1e-5
        close(pd_, P["w"], 1e-5, "adam step %d" % t)
    assert int(step) == 3 and int(seedt) == 3
    # a set cluster error word (sticky hand-off-timeout flag, include/satt_hip.h) makes the update a no-op on the device:
    # garbage gradients - here NaN - never reach parameters or moments
    err = torch.zeros(16, dtype=torch.int32, device=DEV); err[5] = 1
    before = (pd_.clone(), md.clone(), vd.clone())
    gbad = torch.full((n,), float("nan"), device=DEV)
    ops.sumsq(gbad, state)
    ops.adam_step(pd_, gbad, md, vd, state, step, seedt, 5e-4, True, 1.0, 0.9, 0.999, 1e-8, 1.0, 1.0,
                  err_words=[err[0:1].data_ptr(), err[5:6].data_ptr()])
    torch.cuda.synchronize()
    assert torch.equal(pd_, before[0]) and torch.equal(md, before[1]) and torch.equal(vd, before[2])
    ops.sumsq(gd, state)            # clear words: the update happens again
    ops.adam_step(pd_, gd, md, vd, state, step, seedt, 5e-4, True, 1.0, 0.9, 0.999, 1e-8, 1.0, 1.0,
                  err_words=[err[0:1].data_ptr(), err[1:2].data_ptr(), err[2:3].data_ptr()])
    torch.cuda.synchronize()
    assert not torch.equal(pd_, before[0])


@pytest.mark.parametrize("H,B,Tn,Cn", [(16, 3, 9, 2), (64, 5, 23, 4), (256, 32, 40, 4)])
@pytest.mark.parametrize("training", [True, False])
def test_lstm_cluster_fwd_bwd(H, B, Tn, Cn, training):
    """LDS-resident multi-workgroup LSTM (granule all-gather per step) == single-sequence oracle."""
    from satt_amd import ops
    from common import bf16_round
    g = np.random.default_rng(H + Tn)
    xg = g.normal(0, 1, (1, B, Tn, 4 * H)).astype(np.float32)
    Wh = bf16_round(g.normal(0, 1.0 / math.sqrt(H), (H, 4 * H)))
    dh = g.normal(0, 1, (B, Tn, H)).astype(np.float32)
    zc, zh, seed = 0.1, 0.15, 77
    xr = torch.tensor(xg[0], dtype=torch.float64, requires_grad=True)
    W = torch.cat([torch.eye(4 * H, dtype=torch.float64), torch.tensor(Wh, dtype=torch.float64)], 0)
    y = torch_ref.zoneout_lstm_seq(xr, W, torch.zeros(4 * H, dtype=torch.float64), H, None, False, zc, zh, training, seed,
                                   (12, 13))
    y.backward(torch.tensor(dh, dtype=torch.float64))
    assert ops.lstm_cluster_size(B, H) >= Cn or Cn == 2
    Whb, WhT = ops.lstm_cluster_pack(torch.tensor(Wh).to(DEV), H, Cn)      # register-order packs of this cluster size
    e = lambda *s: torch.full(s, 9.0, device=DEV)
    hout, gates, cn, cs, hs = e(B * Tn, H), e(1, B * Tn, 4 * H), e(1, B * Tn, H), e(1, B * Tn, H), e(1, B * Tn, H)
    seedt = torch.tensor([seed], dtype=torch.int32, device=DEV)
    ws = ops.lstm_cluster_ws(B, H, Cn, DEV)
    ops.lstm_cluster_fwd(T(xg), Whb, B, Tn, H, Cn, training, zc, zh, seedt, 12, 13, hout, gates, cn, cs, hs, ws)
    ops.lstm_cluster_status(ws, B, H, Cn)
    close(hout.view(B, Tn, H), y, 1e-5, "cluster lstm fwd")
    dxg = e(1, B * Tn, 4 * H)
    ops.lstm_cluster_bwd(T(dh).view(B * Tn, H), WhT, B, Tn, H, Cn, training, zc, zh, seedt, 12, 13, gates, cn, cs, dxg, ws)
    ops.lstm_cluster_status(ws, B, H, Cn)
    close(dxg.view(B, Tn, 4 * H), xr.grad, 5e-5, "cluster lstm dxg")


def test_lstm_cluster_chunked_pass_equals_full_range():
    """A pass split into chunk launches on ONE workspace (zeroed by the first launch only: step tags and the placement
    handshake tag are unique per launch) gives the same states / gradients as the full-range launch."""
    from satt_amd import ops
    H, B, Tn, Cn = 64, 5, 23, 4
    g = np.random.default_rng(5)
    xg = torch.tensor(g.normal(0, 1, (1, B * Tn, 4 * H)).astype(np.float32)).to(DEV)
    Wh = torch.tensor(g.normal(0, 1.0 / math.sqrt(H), (H, 4 * H)).astype(np.float32)).to(DEV)
    dh = torch.tensor(g.normal(0, 1, (B * Tn, H)).astype(np.float32)).to(DEV)
    pf, pb = ops.lstm_cluster_pack(Wh, H, Cn)
    seedt = torch.tensor([3], dtype=torch.int32, device=DEV)
    e = lambda *s: torch.full(s, 9.0, device=DEV)

    def run(chunks):
        hout, gates, cn, cs, hs = e(B * Tn, H), e(1, B * Tn, 4 * H), e(1, B * Tn, H), e(1, B * Tn, H), e(1, B * Tn, H)
        ws = ops.lstm_cluster_ws(B, H, Cn, DEV)
        ws[:-64] = 0x5A          # garbage granules: the first launch must zero them (the 64-byte status tail is the owner's)
        for (t0, t1) in chunks:
            ops.lstm_cluster_fwd(xg, pf, B, Tn, H, Cn, True, 0.1, 0.15, seedt, 12, 13, hout, gates, cn, cs, hs, ws, t0, t1)
        ops.lstm_cluster_status(ws, B, H, Cn)
        dxg, bst = e(1, B * Tn, 4 * H), e(B, 2, H)
        for (t0, t1) in reversed(chunks):
            ops.lstm_cluster_bwd(dh, pb, B, Tn, H, Cn, True, 0.1, 0.15, seedt, 12, 13, gates, cn, cs, dxg, ws, t0, t1, bst)
        ops.lstm_cluster_status(ws, B, H, Cn)
        return hout, dxg

    h_full, d_full = run([(0, Tn)])
    h_chk, d_chk = run([(0, 5), (5, 6), (6, 17), (17, Tn)])
    assert torch.equal(h_full, h_chk) and torch.equal(d_full, d_chk)


def test_stream_concurrency_probe():
    """the guard of the single-launch attention backward: work of a second stream must progress while a kernel of the
    first one is running; a stream 'paired' with itself is the canonical shared-queue case and must be reported"""
    from satt_amd import ops
    from satt_amd.engine import Engine
    main = torch.cuda.current_stream()
    s1, s2, s3 = Engine._device_streams(torch.device(DEV))
    assert ops.streams_run_concurrently(main, s1) and ops.streams_run_concurrently(main, s2)
    assert not ops.streams_run_concurrently(s3, s3)


@pytest.mark.parametrize("B,T,rate", [(3, 37, 0.0), (2, 160, 0.05), (1, 300, 0.3)])
def test_small_attn_matches_the_unfused_path_and_fp64(B, T, rate):
    """csrc/small_attn.hip (head depth 16: the encoder block) against (i) float64 torch with the same counter-based dropout mask
    and (ii) its own backward by autograd of that reference; the mask index is the one of satt_softmax_fwd"""
    from satt_amd import ops
    from oracle import rng
    heads, hd = 2, 16
    D = heads * hd
    g = np.random.default_rng(T)
    kvq = torch.tensor(g.normal(0, 1.0, (B * T, 3 * D)).astype(np.float32)).to(DEV)
    do = torch.tensor(g.normal(0, 1.0, (B * T, D)).astype(np.float32)).to(DEV)
    seedt = torch.tensor([11], dtype=torch.int32, device=DEV)
    drop = ops.Drop(rate, 7, seedt)
    scale = 1.0 / math.sqrt(hd)
    assert ops.small_attn_supported(hd, T) and not ops.small_attn_supported(8, T)
    p = torch.full((B * heads, T, T), 9.0, device=DEV); o = torch.full((B * T, D), 9.0, device=DEV)
    ops.small_attn_fwd(kvq, D, p, o, B, T, heads, scale, drop)
    dkvq = torch.full((B * T, 3 * D), 9.0, device=DEV); rs = torch.empty(B * heads, T, device=DEV)
    ops.small_attn_bwd(kvq, D, p, do, dkvq, rs, B, T, heads, scale, drop)
    # float64 reference
    x = kvq.double().cpu().requires_grad_(True)
    K, V, Q = (x[:, i * D:(i + 1) * D].reshape(B, T, heads, hd).permute(0, 2, 1, 3) for i in range(3))
    P = torch.softmax(Q @ K.transpose(-1, -2) * scale, -1)
    if drop.thresh:
        keep = torch.tensor(rng.keep_mask(11, 7, (B, heads, T, T), rate))
        Pd = torch.where(keep, P * float(drop.scale), torch.zeros_like(P))
    else:
        Pd = P
    O = (Pd @ V).permute(0, 2, 1, 3).reshape(B * T, D)
    O.backward(do.double().cpu())
    close(p.view(B, heads, T, T), P, 2e-6, "small attn probabilities")
    close(o, O, 1e-5, "small attn output")
    close(dkvq, x.grad, 2e-5, "small attn dK|dV|dQ")


@pytest.mark.parametrize("rows,nl", [(5120, 4), (77, 4), (32, 1), (1000, 3)])
def test_highway_stack_one_launch(rows, nl):
    """csrc/highway.hip: all layers of the CBHG highway stack in one launch per direction against (a) the per-layer form it
    replaces (bf16 GEMM + gate kernel: same operand rounding, differences = summation order, compounding over the layers) and
    (b) float64 of the same function (modules/module.py:258-277) on bf16-rounded weights"""
    from satt_amd import ops
    ops.set_precision("bf16")
    H = 128
    g = np.random.default_rng(5 + rows)
    x = T(g.normal(0, 1.0, (rows, H)))
    Ws, bs = [], []
    for n in range(nl):
        w = T(g.normal(0, 1.0 / math.sqrt(H), (H, 2 * H)))
        Ws.append(ops.Weight(w, w.t().contiguous().to(torch.bfloat16), w.to(torch.bfloat16)))
        bs.append(T(np.concatenate([g.normal(0, 0.1, H), g.normal(-1.0, 0.1, H)])))
    dy = T(g.normal(0, 1.0, (rows, H)))
    assert ops.highway_stack_ok(Ws, H)
    new = lambda *s: torch.full(s, float("nan"), device=DEV)
    zs, ys = [new(rows, 2 * H) for _ in Ws], [new(rows, H) for _ in Ws]
    ops.highway_stack_fwd(x, Ws, bs, zs, ys)
    dzs, dx = [new(rows, 2 * H) for _ in Ws], new(rows, H)
    ops.highway_stack_bwd(dy, x, Ws, zs, ys, dzs, dx)
    # (a) per-layer form
    hws, zr = [x], []
    for n in range(nl):
        z = new(rows, 2 * H); ops.linear(hws[-1], Ws[n], bs[n], z)
        y = new(rows, H); ops.highway_fwd(z, hws[-1], y)
        zr.append(z); hws.append(y)
    d = dy
    dzr = [None] * nl
    for n in reversed(range(nl)):
        dz, dxd = new(rows, 2 * H), new(rows, H)
        ops.highway_bwd(d, zr[n], hws[n], dz, dxd)
        ops.linear_dx(dz, Ws[n], dxd, accumulate=True)
        dzr[n] = dz; d = dxd
    torch.cuda.synchronize()
    for n in range(nl):
        close(zs[n], zr[n], 2e-3, "z%d vs per-layer" % n); close(ys[n], hws[n + 1], 2e-3, "y%d vs per-layer" % n)
        close(dzs[n], dzr[n], 4e-3, "dz%d vs per-layer" % n)
    close(dx, d, 4e-3, "dx vs per-layer")
    # (b) float64 on the rounded weights (activations unrounded: bf16-operand error level)
    xd = x.double().cpu().requires_grad_(True)
    h = xd
    for n in range(nl):
        w = Ws[n].n.double().cpu()
        z = h @ w + bs[n].double().cpu()
        t = torch.sigmoid(z[:, H:])
        h = torch.relu(z[:, :H]) * t + h * (1 - t)
    h.backward(dy.double().cpu())
    close(ys[-1], h, 2e-2, "stack output vs f64")
    # the gradient by relative L2: bf16 rounding flips a few ReLU decisions of near-zero pre-activations (isolated elements
    # change by their whole value), which a max-norm comparison would report as 10 %
    b_ = xd.grad.ravel()
    l2 = float((dx.double().cpu().ravel() - b_).norm() / b_.norm())
    l2_ref = float((d.double().cpu().ravel() - b_).norm() / b_.norm())       # the per-layer form's distance: the same arithmetic
    print("stack dx vs f64: rel L2 %.3e (per-layer form: %.3e)" % (l2, l2_ref))
    assert l2 < 5e-2 and l2 < 1.2 * l2_ref + 1e-4, (l2, l2_ref)
