#!/usr/bin/env python
"""Stand-alone time of the encoder's bidirectional ZoneoutLSTM kernels (register-resident MFMA form, H = 128) at the benchmark
shape: us per launch and per step.  SATT_LIB_PATH selects the library (A/B against tools/probes/libsatt_base.so)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satt_amd  # noqa: F401
from satt_amd import ops
ops.set_precision("bf16")
B, T, H, nd = 32, 160, 128, 2
dev = "cuda"
xg = torch.randn(nd, B * T, 4 * H, device=dev)
Wh = (torch.randn(nd, H, 4 * H, device=dev) / H ** 0.5).to(torch.bfloat16).contiguous()
WhT = Wh.transpose(1, 2).contiguous()
lens = torch.full((B,), T, dtype=torch.int64, device=dev)
seed = torch.tensor([5], dtype=torch.int32, device=dev)
hout = torch.empty(B * T, nd * H, device=dev)
e = lambda *s: torch.empty(*s, device=dev)
gates, cn, cs, hs = e(nd, B * T, 4 * H), e(nd, B * T, H), e(nd, B * T, H), e(nd, B * T, H)
dh, dxg = torch.randn(B * T, nd * H, device=dev), e(nd, B * T, 4 * H)


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


f = t(lambda: ops.lstm_fwd(xg, Wh, lens, nd, B, T, H, True, 0.1, 0.1, seed, (3, 5), (4, 6), hout, gates, cn, cs, hs))
b = t(lambda: ops.lstm_bwd(dh, WhT, lens, nd, B, T, H, True, 0.1, 0.1, seed, (3, 5), (4, 6), gates, cn, cs, dxg))
print("%s: fwd %.1f us (%.3f us/step)  bwd %.1f us (%.3f us/step)" % (os.environ.get("SATT_LIB_PATH", "in-tree"), f, f / T, b, b / T))
from satt_amd import _lib
import ctypes
if hasattr(_lib.lib(), "satt_lstm_prof_read"):      # a -DSATT_LSTM_PROF variant: s_memtime sums per phase of one forward launch
    ops.lstm_fwd(xg, Wh, lens, nd, B, T, H, True, 0.1, 0.1, seed, (3, 5), (4, 6), hout, gates, cn, cs, hs)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    _lib.lib().satt_lstm_prof_read(buf)
    v = [x / T for x in buf]
    print("  encoder LSTM forward, s_memtime ticks per step: top->mfma-start %.1f  mfma %.1f  cell+stores %.1f  issue %.1f  barrier %.1f  (sum %.1f)"
          % (tuple(v[:5]) + (sum(v[:5]),)))

# the decoder's cluster LSTM (H = 256, 4 workgroups per sample) on pipeline-chunk lengths: us per launch -> slope / intercept
D, Cn = 256, ops.lstm_cluster_size(B, 256)
Td = 400
Wc = torch.randn(D, 4 * D, device=dev) / D ** 0.5
pf, pb = ops.lstm_cluster_pack(Wc, D, Cn)
xgc = torch.randn(1, B * Td, 4 * D, device=dev)
houtc, gc, cnc, csc, hsc = e(B * Td, D), e(1, B * Td, 4 * D), e(1, B * Td, D), e(1, B * Td, D), e(1, B * Td, D)
dhc, dxgc, bst = torch.randn(B * Td, D, device=dev), e(1, B * Td, 4 * D), e(B, 2, D)
ws = ops.lstm_cluster_ws(B, D, Cn, dev)
ops.lstm_cluster_fwd(xgc, pf, B, Td, D, Cn, True, 0.1, 0.1, seed, 12, 13, houtc, gc, cnc, csc, hsc, ws, 0, Td)
res = []
for n in (8, 16, 32, 64, 128):
    f = t(lambda: ops.lstm_cluster_fwd(xgc, pf, B, Td, D, Cn, True, 0.1, 0.1, seed, 12, 13, houtc, gc, cnc, csc, hsc, ws, 100, 100 + n))
    bb = t(lambda: ops.lstm_cluster_bwd(dhc, pb, B, Td, D, Cn, True, 0.1, 0.1, seed, 12, 13, gc, cnc, csc, dxgc, ws, 100, 100 + n, bst))
    res.append((n, f, bb))
ops.lstm_cluster_status(ws, B, D, Cn)
print("cluster LSTM (C=%d) us per launch by chunk length: " % Cn + "  ".join("%d: fwd %.1f bwd %.1f" % r for r in res))
(n0, f0, b0), (n1, f1, b1) = res[1], res[-1]
print("  slope fwd %.2f bwd %.2f us/step; intercept fwd %.1f bwd %.1f us" % ((f1 - f0) / (n1 - n0), (b1 - b0) / (n1 - n0),
      f0 - n0 * (f1 - f0) / (n1 - n0), b0 - n0 * (b1 - b0) / (n1 - n0)))

# fused input projection (K = 544 -> 4 * 256) inside the cluster launch against the separate GEMM launch, per chunk length
Kin = 544
xin = torch.randn(B * Td, Kin, device=dev)
Wfull = torch.randn(Kin + D, 4 * D, device=dev) / Kin ** 0.5
bias = torch.randn(4 * D, device=dev)
st = torch.zeros(Wfull.numel(), dtype=torch.bfloat16, device=dev); sn = torch.zeros_like(st)
ops.shadow_pack(Wfull, torch.tensor([0, 1, Kin + D, 4 * D], dtype=torch.int64, device=dev), 1, st, sn)
Wobj = ops.Weight(Wfull, st.view(4 * D, Kin + D), sn.view(Kin + D, 4 * D))
pin = ops.lstm_cluster_pack_in(Wfull[:Kin], D, Cn)
for n in (8, 16, 32):
    def sep():
        ops.linear_rows(xin, Wobj.rows(0, Kin), bias, xgc[0], B, Td, 100, 100 + n)
        ops.lstm_cluster_fwd(xgc, pf, B, Td, D, Cn, True, 0.1, 0.1, seed, 12, 13, houtc, gc, cnc, csc, hsc, ws, 100, 100 + n)
    fz = lambda: ops.lstm_cluster_fwd_x(xin, Kin, pin, bias, xgc, pf, B, Td, D, Cn, True, 0.1, 0.1, seed, 12, 13, houtc, gc, cnc, csc, hsc, ws, 100, 100 + n)
    print("chunk %d steps: GEMM launch + LSTM launch %.1f us, fused launch %.1f us" % (n, t(sep), t(fz)))
