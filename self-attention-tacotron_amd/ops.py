"""Tensor-level wrappers over the C-ABI (include/satt_hip.h).  torch is used only for device memory and the
current HIP stream; every arithmetic op below runs in libsatt_hip.so.  No fallback paths."""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID, ACT_SOFTSIGN, PREC_BF16, PREC_F32, GemmParams

_state = {"prec": PREC_BF16}


def set_precision(name):
    """'bf16' (MFMA bf16 operands, fp32 accumulate — the benchmark dtype) or 'f32' (exact fp32 MFMA; parity mode)."""
    _state["prec"] = {"bf16": PREC_BF16, "f32": PREC_F32}[name]


def get_precision():
    return "bf16" if _state["prec"] == PREC_BF16 else "f32"


def _p(t):
    return None if t is None else t.data_ptr()


def bind_device(device):
    """the device whose current stream the launches go to (one process drives one GPU: Engine / DecodeSession call this once).
    torch.cuda.current_stream() without an index resolves the device through five Python layers incl. an os.getenv per call -
    at ~380 calls per train step that was a third of the host's enqueue time."""
    idx = torch.device(device).index
    _state["dev"] = torch.cuda.current_device() if idx is None else idx


def _dev_index():
    d = _state.get("dev")
    if d is None:
        d = _state["dev"] = torch.cuda.current_device()
    return d


def current_stream():
    """torch's current stream object on the bound device (cheap form of torch.cuda.current_stream())"""
    return torch.cuda.current_stream(_dev_index())


def _s():
    return torch._C._cuda_getCurrentRawStream(_dev_index())


class on_stream:
    """`with ops.on_stream(s):` - torch.cuda.stream(s) without resolving the device on every entry and exit (the engine switches
    streams ~70 times per step; torch's context manager spent ~8 us per switch in Python)"""
    __slots__ = ("s", "prev")

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        self.prev = torch._C._cuda_getCurrentStream(_dev_index())       # (stream id, device index, device type)
        s = self.s
        torch._C._cuda_setStream(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)
        return s

    def __exit__(self, *exc):
        p = self.prev
        torch._C._cuda_setStream(stream_id=p[0], device_index=p[1], device_type=p[2])
        return False


def _ld(t):
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), "need a row-major 2-D view"
    return t.stride(0)


def rate_thresh(rate):
    """(uint32 threshold, scale) for a dropout / zoneout rate; (0, 1) disables."""
    if rate <= 0.0:
        return 0, 1.0
    return min(int(rate * 4294967296.0), 0xFFFFFFFF), 1.0 / (1.0 - rate)


class Drop:
    """Dropout descriptor for the GEMM epilogue / softmax: rate, stream id, device seed tensor (uint32 as int32)."""

    def __init__(self, rate, stream, seed):
        self.thresh, self.scale = rate_thresh(rate)
        self.stream = stream
        self.seed = seed


class Weight:
    """GEMM weight operand: the fp32 master view `w` ([K, N] or [taps, Cin, Cout]) plus its bf16 shadows (satt_shadow_pack):
    `t` = per-tap transpose ([N, K] / [taps, Cout, Cin]) consumed by forward products, `n` = plain cast consumed by input
    gradients.  Every op below also accepts a bare fp32 tensor (no shadows: generic kernel)."""
    __slots__ = ("w", "t", "n")

    def __init__(self, w, t=None, n=None):
        self.w, self.t, self.n = w, t, n

    @property
    def shape(self):
        return self.w.shape

    def rows(self, r0, r1):
        """the weight of the input features [r0, r1) of a Dense layer: W[r0:r1, :]"""
        return Weight(self.w[r0:r1], None if self.t is None else self.t[:, r0:r1],
                      None if self.n is None else self.n[r0:r1])


def _wsplit(W):
    return (W.w, W.t, W.n) if isinstance(W, Weight) else (W, None, None)


def gemm(M, N, K, A, lda, B, sb_k, sb_n, Cm, ldc, *, a_mode=0, conv=None, kin=0, sb_tap=0, batch=(1, 1),
         sA=(0, 0), sB=(0, 0), sC=(0, 0), bias=None, act=ACT_NONE, alpha=1.0, accumulate=False, splitk=1,
         residual=None, ldr=0, drop=None, prec=None, bank=None, Bs=None, sbs_tap=0, sbs_n=0, colsum=None,
         split_overwrite=False, only_path=None):
    """split_overwrite: a split-K product that REPLACES C (not +=): with a slab workspace the kernels write C directly,
    otherwise C is zeroed here and the splits add atomically.  only_path: launch only if satt_gemm_path says so (returns
    False otherwise, nothing launched)."""
    p = GemmParams()
    p.M, p.N, p.K = M, N, K
    p.nb_outer, p.nb_inner = batch
    p.A, p.lda, p.strideA_o, p.strideA_i, p.a_mode = _p(A), lda, sA[0], sA[1], a_mode
    if conv is not None:
        p.conv_T, p.conv_C, p.conv_sgn, p.conv_off = conv
    p.B, p.sb_tap, p.sb_k, p.sb_n, p.strideB_o, p.strideB_i, p.kin = _p(B), sb_tap, sb_k, sb_n, sB[0], sB[1], kin
    p.C, p.ldc, p.strideC_o, p.strideC_i = _p(Cm), ldc, sC[0], sC[1]
    p.bias = _p(bias)
    p.residual, p.ldr = _p(residual), ldr
    p.act, p.alpha, p.accumulate, p.splitk = act, alpha, int(accumulate), splitk
    if drop is not None and drop.thresh:
        p.drop_thresh, p.drop_scale, p.drop_stream, p.seed = drop.thresh, drop.scale, drop.stream, _p(drop.seed)
    p.precision = _state["prec"] if prec is None else prec
    if bank is not None:
        p.bank_ng, p.bank_a_col, p.bank_c_col, p.bank_b_unit = bank
    if Bs is not None:
        p.Bs, p.sbs_tap, p.sbs_n = _p(Bs), sbs_tap, sbs_n
    p.colsum = _p(colsum)
    l = _lib.lib()
    if only_path is not None and l.satt_gemm_path(C.byref(p)) != only_path:
        return False
    # split reductions: a slab workspace (caller-owned: here, torch's caching allocator on the current stream) replaces
    # the fp32 atomics; the tensor must stay referenced until the launch has been issued
    ws = None
    nws = l.satt_gemm_ws_floats(C.byref(p))
    if nws > 0:
        ws = _slab_ws(nws, Cm.device)
        p.ws = ws.data_ptr()
    if split_overwrite and splitk > 1:
        if ws is None:
            Cm.zero_()
            p.accumulate = 1
        else:
            p.accumulate = 0
    if gemm_path_log is not None:
        gemm_path_log.append(l.satt_gemm_path(C.byref(p)))
    _lib.check(l.satt_gemm(C.byref(p), _s()), "satt_gemm")
    return True


gemm_path_log = None    # tests: set to [] to record the kernel family (satt_gemm_path) of every GEMM call


_ws_cache = {}


def _slab_ws(n, device):
    """slab workspace of the CURRENT stream (grow-only, one per stream): a split GEMM and its slab reduction run back to
    back on one stream, so launches of the same stream can share the buffer - and no allocator traffic (a first-time
    (stream, size) request of the caching allocator is a synchronising hipMalloc in the middle of a step)"""
    key = (str(device), _s())
    t = _ws_cache.get(key)
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 1 << 22), dtype=torch.float32, device=device)
        _ws_cache[key] = t
    return t


def _splitk(tiles, k, target=320):
    """number of reduction splits: generic 64x64 kernel - enough workgroups to fill the chip; large tiles (benchmark
    precision) - about `target` workgroups of at least 16 K steps each"""
    if _state["prec"] != PREC_BF16:
        want = max(1, 512 // max(tiles, 1))
        return max(1, min(want, (k + 255) // 256, 64))
    want = max(1, -(-target // max(tiles, 1)))
    # at least 10 K steps of 32 rows per workgroup, at most 32 slices: the small weight-gradient products (1 - 12 output tiles) are
    # bound by the per-step load latency of their workgroups, not by the slab traffic (tools/probes/dw_splitk_sweep.py:
    # 128 x 256 x 5120 takes 23.6 us in 8 slices, 18.7 in 16; 256 x 256 x 12800 27.9 in 24, 27.4 in 32, 32.7 in 64)
    sk = max(1, min(want, k // 320, 32))
    # multiples of 8 let the weight-gradient kernel give every XCD whole reduction slices (csrc/gemm_tile.hip, gemm_dw_k: the
    # operands of a slice then cross the fabric once instead of once per tile row / column)
    if sk >= 8:
        sk = (sk + 3) // 8 * 8 if (sk + 3) // 8 * 8 <= max(8, min(k // 320, 32)) else sk // 8 * 8
    return sk


def _tiles(m, n, bm=128, bn=128):
    """output tiles of an [m, n] product: bm x bn on the large-tile kernels (benchmark precision), 64 x 64 otherwise"""
    if _state["prec"] != PREC_BF16:
        bm = bn = 64
    return ((m + bm - 1) // bm) * ((n + bn - 1) // bn)


def linear(x, W, b, out, act=ACT_NONE, drop=None, residual=None, accumulate=False):
    """out[M,N] = act(x[M,K] @ W[K,N] + b) (dropout) (+residual).  All 2-D row-major views."""
    M, K = x.shape
    W, Wt, _ = _wsplit(W)
    N = W.shape[1]
    gemm(M, N, K, x, _ld(x), W, _ld(W), 1, out, _ld(out), bias=b, act=act, drop=drop, residual=residual,
         ldr=_ld(residual) if residual is not None else 0, accumulate=accumulate,
         Bs=Wt, sbs_n=Wt.stride(0) if Wt is not None else 0)


def linear_rows(x, W, b, out, B, T, t0, t1):
    """out[b, t0:t1, :] = x[b, t0:t1, :] @ W + b for every sample b (row subset of [B*T, *] matrices)."""
    W, Wt, _ = _wsplit(W)
    K, N = x.shape[1], W.shape[1]
    ldx, ldo = _ld(x), _ld(out)
    gemm(t1 - t0, N, K, x[t0:], ldx, W, _ld(W), 1, out[t0:], ldo, bias=b, batch=(B, 1), sA=(T * ldx, 0),
         sC=(T * ldo, 0), Bs=Wt, sbs_n=Wt.stride(0) if Wt is not None else 0)


def linear_dx_rows(dy, W, dx, B, T, t0, t1, accumulate=False):
    """dx[b, t0:t1, :] (+)= dy[b, t0:t1, :] @ W^T for every sample b."""
    W, _, Wn = _wsplit(W)
    N, K = dy.shape[1], W.shape[0]
    ldy, ldx = _ld(dy), _ld(dx)
    gemm(t1 - t0, K, N, dy[t0:], ldy, W, 1, _ld(W), dx[t0:], ldx, batch=(B, 1), sA=(T * ldy, 0), sC=(T * ldx, 0),
         accumulate=accumulate, Bs=Wn, sbs_n=Wn.stride(0) if Wn is not None else 0)


def linear_dx(dy, W, dx, accumulate=False, residual=None):
    """dx[M,K] (+)= dy[M,N] @ W[K,N]^T (+ residual[M,K])"""
    M, N = dy.shape
    W, _, Wn = _wsplit(W)
    K = W.shape[0]
    gemm(M, K, N, dy, _ld(dy), W, 1, _ld(W), dx, _ld(dx), accumulate=accumulate, residual=residual,
         ldr=_ld(residual) if residual is not None else 0, Bs=Wn, sbs_n=Wn.stride(0) if Wn is not None else 0)


def linear_dw(x, dy, dW, db=None):
    """dW[K,N] += x[M,K]^T @ dy[M,N]   (split-K, atomic accumulate); db[N] += column sums of dy (bias gradient)"""
    M, K = x.shape
    N = dy.shape[1]
    gemm(K, N, M, x, _ld(x), dy, _ld(dy), 1, dW, _ld(dW), a_mode=1, accumulate=True, splitk=_splitk(_tiles(K, N), M),
         colsum=db)


def conv1d(x, T, W, out):
    """SAME Conv1D over time: x [B*T, Cin] rows (b,t); W [k,Cin,Cout] contiguous; out [B*T, Cout] view.
    Few output tiles with a long reduction (the 2048-channel projection conv) are split along K."""
    M, Cin = x.shape
    W, Wt, _ = _wsplit(W)
    k, _, Cout = W.shape
    tiles = _tiles(M, Cout, 64, 128) if Wt is not None else ((M + 63) // 64) * ((Cout + 63) // 64)
    # (480 workgroups: the 2048-channel projection conv at 5120 rows, 80 tiles x 6144 - 4 slices 40.1 us, 6 slices 30.2, 8 33.8:
    # tools/probes/proj1_splitk_sweep.py)
    sk = _splitk(tiles, k * Cin, target=480) if (tiles < 256 and k * Cin >= 2048 and out.is_contiguous()) else 1
    gemm(M, Cout, k * Cin, x, _ld(x), W, Cout, 1, out, _ld(out), a_mode=2, conv=(T, Cin, 1, -((k - 1) // 2)),
         kin=Cin, sb_tap=Cin * Cout, splitk=sk, split_overwrite=True, Bs=Wt, sbs_tap=Cin * Cout, sbs_n=Cin)


def conv1d_dx(dy, T, W, dx, accumulate=False):
    """dx[B*T,Cin] (+)= conv-transpose of dy[B*T,Cout] with W[k,Cin,Cout]."""
    M, Cout = dy.shape
    W, _, Wn = _wsplit(W)
    k, Cin, _ = W.shape
    gemm(M, Cin, k * Cout, dy, _ld(dy), W, 1, Cout, dx, _ld(dx), a_mode=2, conv=(T, Cout, -1, (k - 1) // 2),
         kin=Cout, sb_tap=Cin * Cout, accumulate=accumulate, Bs=Wn, sbs_tap=Cin * Cout, sbs_n=Cout)


def conv_bank(x, T, Wall, ng, out):
    """out[:, g*Cout:(g+1)*Cout] = SAME conv of width g+1 over x, g = 0..ng-1, in ONE launch.  Wall: the weights
    [1,Cin,Cout], [2,Cin,Cout], ... [ng,Cin,Cout] contiguous in memory (a flat view); out [B*T, ng*Cout]."""
    M, Cin = x.shape
    Wall, Wt, _ = _wsplit(Wall)
    Cout = out.shape[1] // ng
    gemm(M, Cout, ng * Cin, x, _ld(x), Wall, Cout, 1, out, _ld(out), a_mode=2, conv=(T, Cin, 1, 0),
         kin=Cin, sb_tap=Cin * Cout, bank=(ng, 0, Cout, Cin * Cout), Bs=Wt, sbs_tap=Cin * Cout, sbs_n=Cin)


def conv_bank_dx(dy, T, Wall, ng, dx):
    """dx[B*T, Cin] += sum over the ng widths of the transposed convs of dy[:, g*Cout:(g+1)*Cout] (one launch)."""
    M = dy.shape[0]
    Wall, _, Wn = _wsplit(Wall)
    Cout = dy.shape[1] // ng
    Cin = dx.shape[1]
    gemm(M, Cin, ng * Cout, dy, _ld(dy), Wall, 1, Cout, dx, _ld(dx), a_mode=2, conv=(T, Cout, -1, 0), kin=Cout,
         sb_tap=Cin * Cout, accumulate=True, bank=(ng, Cout, 0, Cin * Cout), Bs=Wn, sbs_tap=Cin * Cout, sbs_n=Cout)


def conv1d_dw(x, T, dy, dW, splitk=None):
    """dW[k,Cin,Cout] += sum over rows of shifted x^T dy."""
    M, Cin = x.shape
    k, _, Cout = dW.shape
    gemm(k * Cin, Cout, M, x, _ld(x), dy, _ld(dy), 1, dW, Cout, a_mode=3, conv=(T, Cin, 1, -((k - 1) // 2)),
         accumulate=True, splitk=_splitk(_tiles(k * Cin, Cout), M) if splitk is None else splitk)


def conv_bank_dw(x, T, dy, dWall, ng):
    """dW of all widths 1..ng of the conv bank: dWall = the gradients [1,Cin,Cout], [2,Cin,Cout], ... contiguous (flat
    view); dy [B*T, ng*Cout].  One launch on the large-tile kernel; per-width calls otherwise."""
    M, Cin = x.shape
    Cout = dy.shape[1] // ng
    tiles = sum(-(-((g + 1) * Cin) // 128) for g in range(ng)) * -(-Cout // 128)
    if gemm(ng * Cin, Cout, M, x, _ld(x), dy, _ld(dy), 1, dWall, Cout, a_mode=3, conv=(T, Cin, 1, 0), accumulate=True,
            splitk=_splitk(tiles, M), bank=(ng, 0, Cout, Cin * Cout), only_path=2):
        return
    off = 0
    for k in range(1, ng + 1):
        n = k * Cin * Cout
        conv1d_dw(x, T, dy[:, (k - 1) * Cout:k * Cout], dWall.view(-1)[off:off + n].view(k, Cin, Cout))
        off += n


def shifted_dw(x, T, shift, dy, dW, db=None):
    """dW[Cx,N] += sum_(b,t) x[b,t+shift,:]^T dy[b,t,:]  (recurrent-weight gradient; zero outside [0,T));
    db[N] += column sums of dy."""
    M, Cx = x.shape
    N = dy.shape[1]
    gemm(Cx, N, M, x, _ld(x), dy, _ld(dy), 1, dW, _ld(dW), a_mode=3, conv=(T, Cx, 1, shift), accumulate=True,
         splitk=_splitk(_tiles(Cx, N), M), colsum=db)


def shadow_pack(flat, table, nweights, st, sn):
    """bf16 shadows (per-tap transpose `st`, plain cast `sn`) of the weights listed in the device table; one launch"""
    _lib.check(_lib.lib().satt_shadow_pack(_p(flat), _p(table), nweights, _p(st), _p(sn), _s()), "shadow_pack")


def embedding_fwd(ids, table, out, offset=0):
    _lib.check(_lib.lib().satt_embedding_fwd(_p(ids), _p(table), _p(out), ids.numel(), table.shape[1], offset, _s()))


def embedding_bwd(ids, dout, dtable, offset=0, atomic=True):
    """dtable[ids - offset] += dout.  atomic=False: deterministic form (one workgroup per table row adds its tokens in ascending
    order): bit-stable, but a padded batch sends a third of its tokens to ONE row (the padding symbol) and that workgroup then
    runs alone - measured 80 us against 37 us for the atomic form on the benchmark batch, so the engine keeps the atomics."""
    if atomic:
        _lib.check(_lib.lib().satt_embedding_bwd(_p(ids), _p(dout), _p(dtable), ids.numel(), dtable.shape[1], offset, _s()))
    else:
        _lib.check(_lib.lib().satt_embedding_bwd_rows(_p(ids), _p(dout), _p(dtable), ids.numel(), dtable.shape[1], offset,
                                                      dtable.shape[0], _s()))


def act_bwd(dy, y, dx, act, scale=1.0):
    rows, cols = y.shape
    _lib.check(_lib.lib().satt_act_bwd(_p(dy), _ld(dy), _p(y), _ld(y), _p(dx), _ld(dx), rows, cols, act, scale, _s()))


def act_bwd_res(dy, z, res, dx, act, scale=1.0):
    """dx = dy * act'(.) with the activation output given as z - res (z = act(u) + res from a GEMM epilogue with a residual)"""
    rows, cols = z.shape
    _lib.check(_lib.lib().satt_act_bwd_res(_p(dy), _ld(dy), _p(z), _ld(z), _p(res), _ld(res), _p(dx), _ld(dx), rows, cols, act,
                                           scale, _s()))


def bn_maxpool_fwd(x, gamma, beta, mp, mean, rstd, mmean, mvar, ws, B, T, eps, momentum, act):
    """mp = maxpool(act(bn(x))) in one pass (x contiguous [B*T, C]); returns False (nothing launched) if the shape does not fit"""
    rc = _lib.lib().satt_bn_maxpool_fwd(_p(x), _p(gamma), _p(beta), _p(mp), _p(mean), _p(rstd), _p(mmean), _p(mvar), _p(ws), B, T,
                                        x.shape[1], eps, momentum, act, _s())
    if rc == -2:
        return False
    _lib.check(rc, "bn_maxpool_fwd")
    return True


def maxpool_bn_bwd(dmp, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, ws, dbuf, B, T, act):
    _lib.check(_lib.lib().satt_maxpool_bn_bwd(_p(dmp), _p(x), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dx), _p(dgamma), _p(dbeta),
                                              _p(ws), _p(dbuf), B, T, x.shape[1], act, _s()), "maxpool_bn_bwd")


def bn_ws(rows, Cc, device):
    return torch.empty(_lib.lib().satt_bn_ws_floats(rows, Cc), dtype=torch.float32, device=device)


def bn_fwd(x, gamma, beta, y, mean, rstd, mmean, mvar, ws, eps, momentum, act):
    rows, Cc = x.shape
    _lib.check(_lib.lib().satt_bn_fwd(_p(x), _ld(x), _p(gamma), _p(beta), _p(y), _ld(y), _p(mean), _p(rstd),
                                      _p(mmean), _p(mvar), _p(ws), rows, Cc, eps, momentum, act, _s()), "bn_fwd")


def bn_fused_state(rows, Cc, device):
    """(ws, sync) of the one-launch BatchNorm pair for one call site: sync is zeroed here ONCE and returned to zero by every
    launch - keep the pair with the call site (engine: one per BatchNorm layer)"""
    l = _lib.lib()
    return (torch.empty(l.satt_bn_fused_ws_floats(rows, Cc), dtype=torch.float32, device=device),
            torch.zeros(l.satt_bn_fused_sync_words(Cc), dtype=torch.int32, device=device))


def bn_fwd_fused(x, gamma, beta, y, mean, rstd, mmean, mvar, state, eps, momentum, act):
    """BatchNorm forward (training statistics) in one launch; False (nothing launched) when the grid is too large for it"""
    rows, Cc = x.shape
    rc = _lib.lib().satt_bn_fwd_fused(_p(x), _ld(x), _p(gamma), _p(beta), _p(y), _ld(y), _p(mean), _p(rstd), _p(mmean), _p(mvar),
                                      _p(state[0]), _p(state[1]), rows, Cc, eps, momentum, act, _s())
    if rc == -2:
        return False
    _lib.check(rc, "bn_fwd_fused")
    return True


def bn_bwd_fused(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, state, act):
    rows, Cc = x.shape
    rc = _lib.lib().satt_bn_bwd_fused(_p(dy), _ld(dy), _p(x), _ld(x), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dx), _ld(dx),
                                      _p(dgamma), _p(dbeta), _p(state[0]), _p(state[1]), rows, Cc, act, _s())
    if rc == -2:
        return False
    _lib.check(rc, "bn_bwd_fused")
    return True


def bn_infer(x, gamma, beta, mmean, mvar, y, eps, act):
    rows, Cc = x.shape
    _lib.check(_lib.lib().satt_bn_infer(_p(x), _ld(x), _p(gamma), _p(beta), _p(mmean), _p(mvar), _p(y), _ld(y), rows,
                                        Cc, eps, act, _s()), "bn_infer")


def bn_bwd(dy, x, gamma, beta, mean, rstd, dx, dgamma, dbeta, ws, act):
    rows, Cc = x.shape
    _lib.check(_lib.lib().satt_bn_bwd(_p(dy), _ld(dy), _p(x), _ld(x), _p(gamma), _p(beta), _p(mean), _p(rstd),
                                      _p(dx), _ld(dx), _p(dgamma), _p(dbeta), _p(ws), rows, Cc, act, _s()), "bn_bwd")


def maxpool_fwd(x, y, B, T, Cc):
    _lib.check(_lib.lib().satt_maxpool_fwd(_p(x), _p(y), B, T, Cc, _s()))


def maxpool_bwd(dy, x, dx, B, T, Cc):
    _lib.check(_lib.lib().satt_maxpool_bwd(_p(dy), _p(x), _p(dx), B, T, Cc, _s()))


def highway_fwd(z, x, y):
    rows, H = x.shape
    _lib.check(_lib.lib().satt_highway_fwd(_p(z), _p(x), _p(y), rows, H, _s()))


def highway_bwd(dy, z, x, dz, dx):
    rows, H = x.shape
    _lib.check(_lib.lib().satt_highway_bwd(_p(dy), _p(z), _p(x), _p(dz), _p(dx), rows, H, _s()))


def _highway_layers(Ws, bs, zs, ys, dzs):
    arr = (_lib.HighwayLayer * len(Ws))()
    for n, W in enumerate(Ws):
        arr[n].Wt, arr[n].Wn = _p(W.t), _p(W.n)
        arr[n].b = _p(bs[n]) if bs is not None else None
        arr[n].z, arr[n].y = _p(zs[n]), _p(ys[n])
        arr[n].dz = _p(dzs[n]) if dzs is not None else None
    return arr


def highway_stack_ok(Ws, H):
    """the one-launch highway stack (csrc/highway.hip) applies: bf16 mode, 128 units, bf16 shadows of every layer"""
    return get_precision() == "bf16" and H == 128 and 0 < len(Ws) <= 8 and \
        all(isinstance(W, Weight) and W.t is not None and W.n is not None for W in Ws)


def highway_stack_fwd(x, Ws, bs, zs, ys):
    """all layers of the highway stack in one launch: zs[n] [rows, 2H] pre-activations, ys[n] [rows, H] outputs of layer n"""
    rows, H = x.shape
    _lib.check(_lib.lib().satt_highway_stack_fwd(_p(x), _highway_layers(Ws, bs, zs, ys, None), len(Ws), rows, H, _s()))


def highway_stack_bwd(dy, x, Ws, zs, ys, dzs, dx):
    """backward of highway_stack_fwd: dzs[n] [rows, 2H] = gradient wrt layer n's pre-activations (the operand of its weight-gradient
    GEMM), dx [rows, H] = gradient wrt the stack input"""
    rows, H = x.shape
    _lib.check(_lib.lib().satt_highway_stack_bwd(_p(dy), _p(x), _highway_layers(Ws, None, zs, ys, dzs), len(Ws), rows, H, _p(dx), _s()))


def colsum(x, out, accumulate=True):
    rows, cols = x.shape
    _lib.check(_lib.lib().satt_colsum(_p(x), _ld(x), _p(out), rows, cols, int(accumulate), _s()))


def loc_filter_dw(a1, dfl, dF, dbF, B, Td, Ti, kernel, filters):
    """location-filter gradients in one launch; False when the dedicated kernel does not serve this filter shape"""
    rc = _lib.lib().satt_loc_filter_dw(_p(a1), _p(dfl), _p(dF), _p(dbF), B, Td, Ti, kernel, filters, _s())
    if rc == -2:
        return False
    _lib.check(rc, "loc_filter_dw")
    return True


def axpby(x, y, a=1.0, b=1.0):
    rows, cols = x.shape
    _lib.check(_lib.lib().satt_axpby(_p(x), _ld(x), _p(y), _ld(y), rows, cols, a, b, _s()))


def seq_mask(x, lengths, y, B, T, Cc, round_bf16=False):
    _lib.check(_lib.lib().satt_seq_mask(_p(x), _p(lengths), _p(y), B, T, Cc, int(round_bf16), _s()))


def bcast_add(sv, y, B, T, Cc):
    _lib.check(_lib.lib().satt_bcast_add(_p(sv), _p(y), B, T, Cc, _s()))


def segment_colsum(x, ds, B, T, Cc, accumulate=False):
    _lib.check(_lib.lib().satt_segment_colsum(_p(x), _p(ds), B, T, Cc, int(accumulate), _s()))


def to_bf16(src, dst, transpose=False):
    rows, cols = src.shape
    _lib.check(_lib.lib().satt_to_bf16(_p(src), _ld(src), _p(dst), rows, cols, int(transpose), _s()))


def softmax_fwd(s, p, pd, nbh, T, scale, causal, drop):
    d = drop if drop is not None else Drop(0.0, 0, None)
    _lib.check(_lib.lib().satt_softmax_fwd(_p(s), _p(p), _p(pd), nbh, T, scale, int(causal), d.thresh, d.scale,
                                           d.stream, _p(d.seed), _s()), "softmax_fwd")


def small_attn_supported(head_dim, T):
    return bool(_lib.lib().satt_small_attn_supported(int(head_dim), int(T)))


def small_attn_fwd(kvq, D, p, o, B, T, heads, scale, drop):
    """fused attention of the encoder block (head depth 16): p [B*H,T,T] probabilities (returned alignments), o [B*T, D]"""
    d = drop if drop is not None else Drop(0.0, 0, None)
    _lib.check(_lib.lib().satt_small_attn_fwd(_p(kvq), _ld(kvq), _p(p), _p(o), _ld(o), B, T, D, heads, scale, d.thresh, d.scale,
                                              d.stream, _p(d.seed), _s()), "small_attn_fwd")


def small_attn_bwd(kvq, D, p, do, dkvq, rowsum, B, T, heads, scale, drop):
    d = drop if drop is not None else Drop(0.0, 0, None)
    _lib.check(_lib.lib().satt_small_attn_bwd(_p(kvq), _ld(kvq), _p(p), _p(do), _ld(do), _p(dkvq), _ld(dkvq), _p(rowsum), B, T, D,
                                              heads, scale, d.thresh, d.scale, d.stream, _p(d.seed), _s()), "small_attn_bwd")


def dropout(x, y, drop):
    """y = dropout(x) with the counter-based mask of `drop` (index = row-major position); also the backward on dy."""
    rows, cols = x.shape
    d = drop if drop is not None else Drop(0.0, 0, None)
    _lib.check(_lib.lib().satt_dropout(_p(x), _ld(x), _p(y), _ld(y), rows, cols, d.thresh, d.scale, d.stream, _p(d.seed),
                                       _s()), "dropout")


def softmax_rows(s, p, rows, cols, scale):
    """p[r, :cols] = softmax(scale * s[r, :cols]); s, p: 2-D views with arbitrary leading dimension."""
    _lib.check(_lib.lib().satt_softmax_rows(_p(s), _ld(s), _p(p), _ld(p), rows, cols, scale, _s()), "softmax_rows")


def softmax_bwd(dpd, p, ds, nbh, T, scale, causal, drop):
    d = drop if drop is not None else Drop(0.0, 0, None)
    _lib.check(_lib.lib().satt_softmax_bwd(_p(dpd), _p(p), _p(ds), nbh, T, scale, int(causal), d.thresh, d.scale,
                                           d.stream, _p(d.seed), _s()), "softmax_bwd")


def flash_attn_supported(hd):
    """the fused attention kernels serve head_dim 128 in the benchmark precision (csrc/flash.hip)"""
    return hd == 128 and _state["prec"] == PREC_BF16


def flash_attn_fwd(kvq, D, o, lse, B, T, H, scale, causal, drop, kvq_b=None):
    """kvq [B*T, 3D] = K | V | Q column blocks (heads of D/H = 128 inside each); o [B*T, D]; lse [B*H, T];
    kvq_b (optional, bf16 [B*T, 3D]): written with the bf16 copies of K | V | Q for the backward kernels"""
    d = drop if drop is not None else Drop(0.0, 0, None)
    if kvq_b is not None:
        _lib.check(_lib.lib().satt_flash_attn_fwd_b(_p(kvq), _p(kvq[:, D:]), _p(kvq[:, 2 * D:]), _ld(kvq), _p(o), _ld(o), _p(lse),
                                                    B, T, H, D // H, scale, int(causal), d.thresh, d.scale, d.stream, _p(d.seed),
                                                    _p(kvq_b), _p(kvq_b[:, D:]), _p(kvq_b[:, 2 * D:]), _ld(kvq_b), _s()), "flash_attn_fwd")
        return
    _lib.check(_lib.lib().satt_flash_attn_fwd(_p(kvq), _p(kvq[:, D:]), _p(kvq[:, 2 * D:]), _ld(kvq), _p(o), _ld(o), _p(lse),
                                              B, T, H, D // H, scale, int(causal), d.thresh, d.scale, d.stream, _p(d.seed),
                                              _s()), "flash_attn_fwd")


FLASH_TILE = 64     # rows per key / query tile of the fused attention kernels (csrc/flash.hip FT)


def flash_attn_bwd(kvq, D, o, do, lse, delta, dkvq, B, T, H, scale, causal, drop, tiles=None, with_delta=True, kvq_b=None, do_b=None):
    """dkvq [B*T, 3D] = dK | dV | dQ (written, not accumulated).  tiles = (lo, hi): only the 64-row tiles [lo, hi) (causal: any
    range leaves its own rows final; ranges run LAST TO FIRST, each with with_delta: a launch computes the row sums o . d o of its own
    query tiles and reads those of later tiles).  kvq_b (bf16 [B*T, 3D] written by
    flash_attn_fwd) + do_b (bf16 [B*T, D] scratch): the bf16-source kernels - bit-identical results, half the operand bytes."""
    d = drop if drop is not None else Drop(0.0, 0, None)
    nt = (T + FLASH_TILE - 1) // FLASH_TILE
    lo, hi = (0, nt) if tiles is None else tiles
    if kvq_b is not None:
        _lib.check(_lib.lib().satt_flash_attn_bwd_tiles_b(
            _p(kvq_b), _p(kvq_b[:, D:]), _p(kvq_b[:, 2 * D:]), _ld(kvq_b), _p(o), _p(do), _ld(o), _p(do_b), _ld(do_b), _p(lse),
            _p(delta), _p(dkvq), _p(dkvq[:, D:]), _p(dkvq[:, 2 * D:]), _ld(dkvq), B, T, H, D // H, scale, int(causal), d.thresh,
            d.scale, d.stream, _p(d.seed), lo, hi, int(with_delta), _s()), "flash_attn_bwd")
        return
    _lib.check(_lib.lib().satt_flash_attn_bwd_tiles(
        _p(kvq), _p(kvq[:, D:]), _p(kvq[:, 2 * D:]), _ld(kvq), _p(o), _p(do), _ld(o), _p(lse), _p(delta), _p(dkvq),
        _p(dkvq[:, D:]), _p(dkvq[:, 2 * D:]), _ld(dkvq), B, T, H, D // H, scale, int(causal), d.thresh, d.scale, d.stream,
        _p(d.seed), lo, hi, int(with_delta), _s()), "flash_attn_bwd")


def _u32arr(vals):
    return (C.c_uint32 * len(vals))(*vals)


def lstm_fwd(xg, Wh, lengths, ndir, B, T, H, training, zc, zh, seed, streams_c, streams_h, hout, gates, cnew,
             cstate, hstate):
    zct, _ = rate_thresh(zc if training else 0.0)
    zht, _ = rate_thresh(zh if training else 0.0)
    _lib.check(_lib.lib().satt_lstm_fwd(_p(xg), _p(Wh), _p(lengths), ndir, B, T, H, int(training), zc, zh, zct, zht,
                                        _p(seed), _u32arr(streams_c), _u32arr(streams_h), _p(hout), _ld(hout),
                                        _p(gates), _p(cnew), _p(cstate), _p(hstate), _s()), "lstm_fwd")


def lstm_bwd(dhout, WhT, lengths, ndir, B, T, H, training, zc, zh, seed, streams_c, streams_h, gates, cnew, cstate,
             dxg):
    zct, _ = rate_thresh(zc if training else 0.0)
    zht, _ = rate_thresh(zh if training else 0.0)
    _lib.check(_lib.lib().satt_lstm_bwd(_p(dhout), _ld(dhout), _p(WhT), _p(lengths), ndir, B, T, H, int(training), zc,
                                        zh, zct, zht, _p(seed), _u32arr(streams_c), _u32arr(streams_h), _p(gates),
                                        _p(cnew), _p(cstate), _p(dxg), _s()), "lstm_bwd")


def lstm_cluster_size(B, H, T=1):
    """workgroups per sample for the register-resident cluster LSTM (0: not applicable -> single-workgroup kernel)."""
    l = _lib.lib()
    for Cn in (4, 8, 2):
        if l.satt_lstm_cluster_check(B, T, H, Cn) == 0:
            rs = [lstm_cluster_residency(B, T, H, Cn, bw) for bw in (False, True)]
            if all(r is None or r[0] <= r[1] * r[2] for r in rs):    # every workgroup of either launch can be resident (None: no device)
                return Cn
    return 0


def lstm_cluster_residency(B, T, H, Cn, backward=False):
    """(workgroups, workgroups per CU, CUs) of a cluster LSTM launch on the current device (satt_lstm_cluster_residency: the
    occupancy calculator's answer for the kernel), or None without a device"""
    n, per, cus = C.c_int(0), C.c_int(0), C.c_int(0)
    if _lib.lib().satt_lstm_cluster_residency(B, T, H, Cn, int(backward), C.byref(n), C.byref(per), C.byref(cus)) != 0:
        return None
    return n.value, per.value, cus.value


def lstm_cluster_ws(B, H, Cn, device):
    """exchange workspace, ZERO-filled: launches clear the granules only, the 64-byte tail (error word, exchange-path
    counters) is sticky until the owner zeroes it again"""
    return torch.zeros(_lib.lib().satt_lstm_cluster_ws_bytes(B, H, Cn), dtype=torch.uint8, device=device)


def cluster_err_word(ws):
    """device pointer of a cluster workspace's sticky error word (adam_step's err arguments)"""
    return ws.data_ptr() + ws.numel() - 64


def lstm_cluster_pack(Wh, H, Cn):
    """register-order bf16 packs (forward, backward) of the fp32 recurrent weights Wh [H, 4H] for cluster size Cn."""
    l = _lib.lib()
    n = l.satt_lstm_cluster_pack_elems(Cn)
    pf = torch.empty(n, dtype=torch.bfloat16, device=Wh.device)
    pb = torch.empty(n, dtype=torch.bfloat16, device=Wh.device)
    _lib.check(l.satt_lstm_cluster_pack(_p(Wh), _ld(Wh), H, Cn, _p(pf), _p(pb), _s()), "lstm_cluster_pack")
    return pf, pb


def lstm_cluster_fwd(xg, Wh, B, T, H, Cn, training, zc, zh, seed, stream_c, stream_h, hout, gates, cnew, cstate, hstate,
                     ws, t0=0, t1=None):
    zct, _ = rate_thresh(zc if training else 0.0)
    zht, _ = rate_thresh(zh if training else 0.0)
    _lib.check(_lib.lib().satt_lstm_cluster_fwd(_p(xg), _p(Wh), B, T, H, Cn, int(training), zc, zh, zct, zht, _p(seed),
                                                stream_c, stream_h, _p(hout), _ld(hout), _p(gates), _p(cnew),
                                                _p(cstate), _p(hstate), _p(ws), t0, T if t1 is None else t1, _s()),
               "lstm_cluster_fwd")


LSTM_CLUSTER_FUSED_KMAX = 544     # input width up to which satt_lstm_cluster_fwd_x forms the input projection itself


def lstm_cluster_pack_in(Win, H, Cn):
    """register-order bf16 pack of the fp32 INPUT weights Win [K, 4H] (a row view of the layer's weight matrix) for the fused input
    projection of lstm_cluster_fwd(x=...), cluster size Cn"""
    l = _lib.lib()
    K = Win.shape[0]
    pk = torch.empty(l.satt_lstm_cluster_pack_in_elems(K, Cn), dtype=torch.bfloat16, device=Win.device)
    _lib.check(l.satt_lstm_cluster_pack_in(_p(Win), _ld(Win), K, H, Cn, _p(pk), _s()), "lstm_cluster_pack_in")
    return pk


def lstm_cluster_fwd_x(x, Kin, Win_pack, bias, xg, Wh, B, T, H, Cn, training, zc, zh, seed, stream_c, stream_h, hout, gates, cnew,
                       cstate, hstate, ws, t0=0, t1=None):
    """lstm_cluster_fwd whose launch first forms xg[:, t0:t1] = x[:, t0:t1, :Kin] Win + bias itself (bit-identical to ops.linear on
    the bf16 shadow; x: fp32 rows [B*T, ldx]) - for the short chunks at the end of the forward pipeline, where a GEMM launch of
    its own costs more than the product"""
    zct, _ = rate_thresh(zc if training else 0.0)
    zht, _ = rate_thresh(zh if training else 0.0)
    _lib.check(_lib.lib().satt_lstm_cluster_fwd_x(_p(xg), _p(Wh), B, T, H, Cn, int(training), zc, zh, zct, zht, _p(seed),
                                                  stream_c, stream_h, _p(hout), _ld(hout), _p(gates), _p(cnew),
                                                  _p(cstate), _p(hstate), _p(ws), t0, T if t1 is None else t1, _p(x), _ld(x),
                                                  int(Kin), _p(Win_pack), _p(bias), _s()),
               "lstm_cluster_fwd_x")


def lstm_cluster_bwd(dhout, WhT, B, T, H, Cn, training, zc, zh, seed, stream_c, stream_h, gates, cnew, cstate, dxg, ws,
                     t0=0, t1=None, bstate=None):
    zct, _ = rate_thresh(zc if training else 0.0)
    zht, _ = rate_thresh(zh if training else 0.0)
    _lib.check(_lib.lib().satt_lstm_cluster_bwd(_p(dhout), _ld(dhout), _p(WhT), B, T, H, Cn, int(training), zc, zh, zct,
                                                zht, _p(seed), stream_c, stream_h, _p(gates), _p(cnew), _p(cstate),
                                                _p(dxg), _p(ws), t0, T if t1 is None else t1, _p(bstate), _s()),
               "lstm_cluster_bwd")


def lstm_cluster_status(ws, B, H, Cn):
    _lib.check(_lib.lib().satt_lstm_cluster_status(_p(ws), B, H, Cn, _s()), "lstm cluster hand-off timeout")


def attn_rnn_params(**kw):
    p = _lib.AttnRnnParams()
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        setattr(p, k, v)
    return p


def attn_rnn_fwd(p):
    _lib.check(_lib.lib().satt_attn_rnn_fwd(C.byref(p), _s()), "attn_rnn_fwd")


def attn_rnn_bwd(fwd_params, **kw):
    pb = _lib.AttnRnnBwdParams()
    pb.f = fwd_params
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        setattr(pb, k, v)
    _lib.check(_lib.lib().satt_attn_rnn_bwd(C.byref(pb), _s()), "attn_rnn_bwd")


ATTN_CLUSTER_SIZES = (4, 8, 2)  # candidate workgroups per sample in order of preference (B * C <= 256; 4 leaves
                                # half of the CUs to the LSTM kernels that run concurrently on the other streams)


def attn_cluster_size(fwd_params):
    """workgroups per sample for the cluster attention kernels (0 -> single-workgroup kernels): the largest candidate
    the library accepts for this problem (satt_attn_cluster_check: sizes, register-resident slice, LDS, residency)."""
    l = _lib.lib()
    for Cn in ATTN_CLUSTER_SIZES:
        if l.satt_attn_cluster_check(C.byref(fwd_params), Cn) == 0:
            rs = [attn_cluster_residency(fwd_params, Cn, bw) for bw in (False, True)]
            if all(r is None or r[0] <= r[1] * r[2] for r in rs):    # every workgroup of either launch can be resident (None: no device)
                return Cn
    return 0


def attn_cluster_residency(fwd_params, Cn, backward=False, vw1=None, **kw):
    """(workgroups, workgroups per CU, CUs) of the launch attn_cluster_fwd / attn_cluster_bwd would make (same kernel selection,
    same LDS size: satt_attn_cluster_residency / _bwd_residency), or None without a device.  kw: fields of the backward
    parameter block that select the kernel (saf comes with fwd_params)."""
    n, per, cus = C.c_int(0), C.c_int(0), C.c_int(0)
    if backward:
        cb = _lib.AttnClusterBwdParams()
        cb.b.f = fwd_params; cb.C = Cn
        rc = _lib.lib().satt_attn_cluster_bwd_residency(C.byref(cb), C.byref(n), C.byref(per), C.byref(cus))
    else:
        cp = _lib.AttnClusterParams()
        cp.f = fwd_params; cp.C = Cn; cp.vw1 = _p(vw1)
        rc = _lib.lib().satt_attn_cluster_residency(C.byref(cp), C.byref(n), C.byref(per), C.byref(cus))
    if rc != 0:
        return None
    return n.value, per.value, cus.value


def attn_cluster_pack(Wrec, A, Cn):
    """per-member bf16 slices of Wrec [K, 4A] (fp32 view): (WrecP in MFMA operand order, WrecTP [C,4A,nwp])."""
    K = Wrec.shape[0]
    l = _lib.lib()
    wp = torch.empty(l.satt_attn_cluster_pack_elems(K, A, Cn, 0), dtype=torch.bfloat16, device=Wrec.device)
    wtp = torch.empty(l.satt_attn_cluster_pack_elems(K, A, Cn, 1), dtype=torch.bfloat16, device=Wrec.device)
    _lib.check(l.satt_attn_cluster_pack(_p(Wrec), _ld(Wrec), _p(wp), _p(wtp), K, A, Cn, _s()), "attn_cluster_pack")
    return wp, wtp


def attn_cluster_ws(fwd_params, Cn, device):
    """exchange workspace, zero-filled (see lstm_cluster_ws)"""
    return torch.zeros(_lib.lib().satt_attn_cluster_ws_bytes(C.byref(fwd_params), Cn), dtype=torch.uint8,
                       device=device)


def attn_cluster_state(fwd_params, Cn, device):
    return torch.empty(_lib.lib().satt_attn_cluster_state_floats(C.byref(fwd_params), Cn), dtype=torch.float32,
                       device=device)


def attn_cluster_fold(fwd_params, Cn):
    """True if the folded form of the forward attention kernel exists for this problem (csrc/attn_cluster.hip, FOLD)"""
    return bool(_lib.lib().satt_attn_cluster_fold(C.byref(fwd_params), Cn))


def attn_cluster_fwd(fwd_params, Cn, WrecP, ws, t0=0, t1=None, progress=None, bounds=(), vw1=None):
    """progress (int32 device tensor, zeroed by the caller) + bounds (chunk end steps): the launch spans several pipeline
    chunks and signals the end of each one - consumers wait with stream_wait_value(progress, (k+1) * B * Cn)"""
    cp = _lib.AttnClusterParams()
    cp.f = fwd_params; cp.C = Cn; cp.WrecP = _p(WrecP); cp.ws = _p(ws)
    cp.t0 = t0; cp.t1 = fwd_params.Td if t1 is None else t1
    cp.vw1 = _p(vw1)
    cp.progress = _p(progress); cp.nbound = len(bounds) if progress is not None else 0
    for i, bnd in enumerate(bounds):
        cp.bound[i] = bnd
    _lib.check(_lib.lib().satt_attn_cluster_fwd(C.byref(cp), _s()), "attn_cluster_fwd")


_hip = None


def _hiprt():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]
        _hip.hipStreamWaitValue32.restype = C.c_int
        _hip.hipStreamWriteValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint]
        _hip.hipStreamWriteValue32.restype = C.c_int
    return _hip


def stream_wait_value(counter, value, stream=None):
    """the stream (default: current) does not run later work until *counter >= value (hipStreamWaitValue32, GTE): the
    producer/consumer edge between a RUNNING persistent kernel that counts its finished chunks and the next layer"""
    st = stream.cuda_stream if stream is not None else _s()
    rc = _hiprt().hipStreamWaitValue32(st, counter.data_ptr(), int(value), 0, 0xFFFFFFFF)
    if rc:
        raise _lib.SattError("hipStreamWaitValue32 failed (%d)" % rc)


def stream_write_value(counter, value, stream=None):
    st = stream.cuda_stream if stream is not None else _s()
    rc = _hiprt().hipStreamWriteValue32(st, counter.data_ptr(), int(value), 0)
    if rc:
        raise _lib.SattError("hipStreamWriteValue32 failed (%d)" % rc)


def attn_cluster_bwd(fwd_params, Cn, WrecTP, ws, t0=0, t1=None, state=None, ready=None, done=None, bounds=(), **kw):
    """ready / done (int32 device words, zeroed by the caller) + bounds (first step of every chunk in processing order,
    i.e. descending): one launch over several pipeline chunks, see satt_attn_cluster_bwd_params"""
    cb = _lib.AttnClusterBwdParams()
    cb.b.f = fwd_params
    cb.t0 = t0; cb.t1 = fwd_params.Td if t1 is None else t1; cb.state = _p(state)
    cb.ready = _p(ready); cb.done = _p(done); cb.nbound = len(bounds) if ready is not None else 0
    for i, bnd in enumerate(bounds):
        cb.bound[i] = bnd
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        setattr(cb.b, k, v)
    cb.C = Cn; cb.WrecTP = _p(WrecTP); cb.ws = _p(ws)
    _lib.check(_lib.lib().satt_attn_cluster_bwd(C.byref(cb), _s()), "attn_cluster_bwd")


def attn_cluster_status(fwd_params, Cn, ws):
    _lib.check(_lib.lib().satt_attn_cluster_status(C.byref(fwd_params), Cn, _p(ws), _s()),
               "attention cluster hand-off timeout")


def attn_cluster_fastpath(fwd_params, Cn, ws):
    """(fast, slow): workgroup-launches on ws since it was zeroed that exchanged through same-XCD plain stores / through
    write-through stores (host-synchronous; tests)"""
    n, m = C.c_int(0), C.c_int(0)
    _lib.check(_lib.lib().satt_attn_cluster_fastpath(C.byref(fwd_params), Cn, _p(ws), _s(), C.byref(n), C.byref(m)), "fastpath")
    return n.value, m.value


def lstm_cluster_fastpath(ws, B, H, Cn):
    n, m = C.c_int(0), C.c_int(0)
    _lib.check(_lib.lib().satt_lstm_cluster_fastpath(_p(ws), B, H, Cn, _s(), C.byref(n), C.byref(m)), "fastpath")
    return n.value, m.value


def attn_param_grads(fwd_params, de1, de2, dkeys1, dkeys2, dv1, db1, dlocU, dv2, t0=None, t1=None, accumulate=False,
                     lds_pad=0):
    """deferred attention gradients of the steps [t0, t1) (default: all); accumulate adds into dkeys1/2."""
    t0 = 0 if t0 is None else t0
    t1 = fwd_params.Td if t1 is None else t1
    _lib.check(_lib.lib().satt_attn_param_grads_range(C.byref(fwd_params), _p(de1), _p(de2), _p(dkeys1), _p(dkeys2),
                                                      _p(dv1), _p(db1), _p(dlocU), _p(dv2), t0, t1, int(accumulate),
                                                      int(lds_pad), _s()), "attn_param_grads")


def attn_param_grads_acc_doubles(fwd_params):
    """float64 elements satt_attn_param_grads_acc needs for this problem (one slot per workgroup of its fixed grid)"""
    return int(_lib.lib().satt_attn_param_grads_acc_doubles(C.byref(fwd_params)))


def attn_param_grads_acc_buffer(fwd_params, device):
    """float64 workgroup slots of the deferred attention gradients (saved-factor path) for THIS problem size (B, Ti)"""
    return torch.zeros(_lib.lib().satt_attn_param_grads_acc_doubles(C.byref(fwd_params)), dtype=torch.float64, device=device)


def attn_param_grads_acc(fwd_params, de1, de2, dkeys1, dkeys2, acc, t0=None, t1=None, accumulate=False, lds_pad=0):
    t0 = 0 if t0 is None else t0
    t1 = fwd_params.Td if t1 is None else t1
    _lib.check(_lib.lib().satt_attn_param_grads_acc(C.byref(fwd_params), _p(de1), _p(de2), _p(dkeys1), _p(dkeys2), _p(acc), t0, t1,
                                                    int(accumulate), int(lds_pad), _s()), "attn_param_grads_acc")


def attn_param_grads_finish(fwd_params, acc, dv1, db1, dlocU, dv2):
    _lib.check(_lib.lib().satt_attn_param_grads_finish(C.byref(fwd_params), _p(acc), _p(dv1), _p(db1), _p(dlocU), _p(dv2), _s()),
               "attn_param_grads_finish")


def loss_fwd_bwd(mel, mel_ld, target, spec_mask, stop, stop_ld, done, bin_mask, B, Tm, nm, Td, l2, losses, dmel,
                 dmel_ld, dstop, dstop_ld, ws):
    _lib.check(_lib.lib().satt_loss_fwd_bwd(_p(mel), mel_ld, _p(target), _p(spec_mask), _p(stop), stop_ld, _p(done),
                                            _p(bin_mask), B, Tm, nm, Td, int(l2), _p(losses), _p(dmel), dmel_ld,
                                            _p(dstop), dstop_ld, _p(ws), _s()), "loss_fwd_bwd")


def loss_mask_sums(spec_mask, bin_mask, B, Tm, Td, ws):
    """mask sums of the batch into ws (8 floats) + reset of its accumulators: call any time before loss_fwd_bwd_presummed"""
    _lib.check(_lib.lib().satt_loss_mask_sums(_p(spec_mask), _p(bin_mask), B, Tm, Td, _p(ws), _s()), "loss_mask_sums")


def loss_fwd_bwd_presummed(mel, mel_ld, target, spec_mask, stop, stop_ld, done, bin_mask, B, Tm, nm, Td, l2, losses, dmel,
                           dmel_ld, dstop, dstop_ld, ws):
    _lib.check(_lib.lib().satt_loss_fwd_bwd_presummed(_p(mel), mel_ld, _p(target), _p(spec_mask), _p(stop), stop_ld, _p(done),
                                                      _p(bin_mask), B, Tm, nm, Td, int(l2), _p(losses), _p(dmel), dmel_ld,
                                                      _p(dstop), dstop_ld, _p(ws), _s()), "loss_fwd_bwd_presummed")


def opt_state(device):
    """optimiser scratch {sumsq, norm, lr_t, scale, per-block partials} for sumsq / adam_step"""
    return torch.zeros(_lib.lib().satt_sumsq_state_floats(), dtype=torch.float32, device=device)


def sumsq(g, state):
    _lib.check(_lib.lib().satt_sumsq(_p(g), g.numel(), _p(state), _s()))


def adam_step(p, g, m, v, state, step_dev, seed_dev, lr0, decay, step_factor, b1, b2, eps, clip, grad_scale, err_words=()):
    """err_words: up to three device pointers (cluster_err_word) - the update is skipped on the device if any is set"""
    e = (list(err_words) + [None, None, None])[:3]
    _lib.check(_lib.lib().satt_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(state), _p(step_dev),
                                         _p(seed_dev), lr0, int(decay), step_factor, b1, b2, eps, clip, grad_scale,
                                         e[0], e[1], e[2], _s()), "adam_step")


def poison_on_error(g, err_words=()):
    """data-parallel guard: g[0] = NaN if any cluster error word is set (call on a bucket's first element before its all-reduce)"""
    e = (list(err_words) + [None, None, None])[:3]
    _lib.check(_lib.lib().satt_poison_on_error(_p(g), e[0], e[1], e[2], _s()), "poison_on_error")


_probe_cache = {}


def streams_run_concurrently(main, other, spins=40000):
    """True if a kernel launched on `other` AFTER a kernel was launched on `main` runs while that kernel is still
    running (the two streams do not share a hardware queue and nothing - e.g. a counter-collecting profiler -
    serialises kernels).  One-off probe per stream pair, ~0.1 ms (60 ms when it fails)."""
    key = (main.cuda_stream, other.cuda_stream)
    if key[0] == key[1]:
        return False
    if key not in _probe_cache:
        buf = torch.zeros(2, dtype=torch.int32, device=main.device)
        torch.cuda.synchronize(main.device)
        _lib.check(_lib.lib().satt_stream_probe(buf[0:1].data_ptr(), buf[1:2].data_ptr(), int(spins), main.cuda_stream,
                                                other.cuda_stream), "stream_probe")
        torch.cuda.synchronize(main.device)
        _probe_cache[key] = bool(int(buf[1]) == 1)
    return _probe_cache[key]


# ---------------------------------------------------------------------- autoregressive decode step (csrc/decode.hip)
def dec_linear_params(xs, W, y, *, bias=None, act=ACT_NONE, res=None, step=None, B=None, lstm=None, step_out=None,
                      stop=None, drop=None, drop_T=0):
    """y = act([x0 | x1 | x2] W + bias) + res.  xs: list of (tensor, features, batch_stride, step_stride[, parity_stride]);
    y / res: (tensor, batch_stride, step_stride); W: fp32 [K, N] view or ops.Weight (bf16 plain-cast shadow in bf16 mode).
    lstm=(H, c_state, h_state, zc, zh): LSTM form - W yields the gates and the ZoneoutLSTMCell runs in the epilogue on the
    [2, B, H] states (double-buffered by step parity); y receives the pre-zoneout cell output.
    step_out=(counter, add): workgroup (0,0) publishes *step + add into `counter` (a word this launch does not read);
    stop=(logits, batch_stride, step_stride, flag, threshold, min_steps): it also evaluates the previous step's stop rule.
    drop (a Drop) / drop_T: dropout that stays on while synthesising (apply_dropout_on_inference) - the training kernels' mask
    over a [B, drop_T, N] activation, row (b, *step).
    Returns the filled parameter block (kept by the caller: the tensors it points to must stay alive)."""
    p = _lib.DecLinearParams()
    w, _, wn = _wsplit(W)
    p.B = int(B)
    p.N = int(w.shape[1])
    p.nseg = len(xs)
    for s, xd in enumerate(xs):
        t, k, bs, ss = xd[:4]
        p.x[s] = t.data_ptr(); p.k[s] = int(k); p.x_bs[s] = int(bs); p.x_ss[s] = int(ss)
        p.x_ps[s] = int(xd[4]) if len(xd) > 4 else 0
    if sum(int(xd[1]) for xd in xs) != w.shape[0]:
        raise _lib.SattError("dec_linear: input segments do not add up to the rows of W")
    if w.dtype == torch.bfloat16:                      # a caller-built bf16 matrix
        p.Wb = w.data_ptr(); p.ldw = int(w.stride(0))
    elif _state["prec"] == PREC_BF16 and wn is not None:
        p.Wb = wn.data_ptr(); p.ldw = int(wn.stride(0))
    else:
        p.W = w.data_ptr(); p.ldw = int(w.stride(0))
    p.bias = _p(bias)
    p.act = int(act)
    if res is not None:
        p.res = res[0].data_ptr(); p.res_bs = int(res[1]); p.res_ss = int(res[2])
    p.y = y[0].data_ptr(); p.y_bs = int(y[1]); p.y_ss = int(y[2])
    p.step = _p(step)
    if step_out is not None:
        p.step_out = step_out[0].data_ptr(); p.step_add = int(step_out[1])
    if stop is not None:
        p.stop = stop[0].data_ptr(); p.stop_bs = int(stop[1]); p.stop_ss = int(stop[2]); p.flag = stop[3].data_ptr()
        p.stop_threshold = float(stop[4]); p.min_steps = int(stop[5])
    if lstm is not None:
        p.lstm_H = int(lstm[0]); p.c_state = lstm[1].data_ptr(); p.h_state = lstm[2].data_ptr()
        p.zc = float(lstm[3]); p.zh = float(lstm[4])
    if drop is not None and drop.thresh:
        p.drop_thresh, p.drop_scale, p.drop_stream, p.drop_seed, p.drop_T = drop.thresh, drop.scale, drop.stream, _p(drop.seed), int(drop_T)
    return p


def dec_linear(p):
    _lib.check(_lib.lib().satt_dec_linear(C.byref(p), _s()), "dec_linear")


def dec_linear2(pa, pb):
    """two plain Dense layers in one launch; returns False (nothing launched) if the pair does not fit the fused kernel"""
    rc = _lib.lib().satt_dec_linear2(C.byref(pa), C.byref(pb), _s())
    if rc == -2:        # SATT_E_UNSUPPORTED
        return False
    _lib.check(rc, "dec_linear2")
    return True


def dec_linear_chain(pre, main):
    """pre (1 or 2 DecLinearParams) -> main in one launch; returns False (nothing launched) if the chain does not fit"""
    arr = (_lib.DecLinearParams * len(pre))(*pre)
    rc = _lib.lib().satt_dec_linear_chain(arr, len(pre), C.byref(main), _s())
    if rc == -2:        # SATT_E_UNSUPPORTED
        return False
    _lib.check(rc, "dec_linear_chain")
    return True


def dec_attention_params(**kw):
    p = _lib.DecAttentionParams()
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        setattr(p, k, v)
    return p


def dec_attention(p):
    _lib.check(_lib.lib().satt_dec_attention(C.byref(p), _s()), "dec_attention")


def dec_mega_params(**kw):
    """parameter block of the persistent decode step (satt_dec_mega_params); tensors are passed as their device pointers"""
    p = _lib.DecMegaParams()
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        setattr(p, k, v)
    return p


def dec_mega_supported(p):
    return bool(_lib.lib().satt_dec_mega_supported(C.byref(p)))


def dec_mega_scratch_floats(B, heads, hd):
    return int(_lib.lib().satt_dec_mega_scratch_floats(B, heads, hd))


def dec_mega(p, nsteps):
    """`nsteps` whole decoder steps in ONE launch (csrc/decode_mega2.hip)"""
    p.nsteps = int(nsteps)
    _lib.check(_lib.lib().satt_dec_mega(C.byref(p), _s()), "dec_mega")


def dec_self_attn(kvq, out, step, B, Td, D, heads, scale):
    _lib.check(_lib.lib().satt_dec_self_attn(_p(kvq), _p(out), _p(step), B, Td, D, heads, float(scale), _s()), "dec_self_attn")


def l2_reg(w, g, table, nseg, scale, reg, total=None):
    """g += scale * w and reg (+ total) += scale * sum(w^2) / 2 over the (offset, count) segments of `table`"""
    _lib.check(_lib.lib().satt_l2_reg(_p(w), _p(g), _p(table), int(nseg), float(scale), _p(reg), _p(total), _s()), "l2_reg")
