"""Per-shape GEMM roofline table of the train step's large products (B=32, Ti=160, Td=400): HIP-event time per launch,
TFLOP/s and fraction of the 2.5 PFLOP/s dense bf16 MFMA peak, ALGORITHMIC bytes (fp32 activations and outputs, bf16 weight
shadows) per launch, GB/s and fraction of the 8 TB/s HBM peak, and which of the two bounds the shape (the larger fraction =
its roofline fraction), large-tile kernels (csrc/gemm_tile.hip) next to the generic 64x64 kernel on the same operands.
`python tools/bench_gemm.py [--iters N]` -> table on stdout."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import satt_amd  # noqa: E402,F401
from satt_amd import ops  # noqa: E402

PEAK = 2500.0
HBM = 8000.0      # GB/s
DEV = "cuda"


def weight(W):
    shp = tuple(W.shape)
    taps, R, Cc = (1,) + shp if len(shp) == 2 else shp
    st = torch.zeros(W.numel(), dtype=torch.bfloat16, device=DEV); sn = torch.zeros_like(st)
    ops.shadow_pack(W, torch.tensor([0, taps, R, Cc], dtype=torch.int64, device=DEV), 1, st, sn)
    return ops.Weight(W, st.view((Cc, R) if len(shp) == 2 else (taps, Cc, R)), sn.view(shp))


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-dw", action="store_true")
    ap.add_argument("--no-generic", action="store_true")
    args = ap.parse_args()
    ops.set_precision("bf16")
    # sustained clocks: ~0.2 s of dense work before the first measurement
    wa, wb = torch.randn(8192, 8192, device=DEV), torch.randn(8192, 8192, device=DEV)
    for _ in range(12):
        wa = (wa @ wb) * 1e-2
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    rows = []

    def case(name, flop, tile_fn, gen_fn, nbytes=0.0):
        if args.no_dw and gen_fn is None:
            return
        t1 = timeit(tile_fn, args.iters)
        t0 = timeit(gen_fn, args.iters) if (gen_fn is not None and not args.no_generic) else float("nan")
        fm, fh = flop / t1 / 1e6 / PEAK, nbytes / t1 / 1e3 / HBM
        rows.append((name, flop / 1e9, t1, flop / t1 / 1e6, fm, nbytes / 1e6, nbytes / t1 / 1e3, fh,
                     "hbm" if fh > fm else "mfma", max(fm, fh), t0, flop / t0 / 1e6 if t0 == t0 else float("nan")))

    # ---- Dense forward / dX / dW
    for (M, N, K, tag) in [(12800, 1024, 128, "xg_att"), (12800, 1024, 544, "lstm1 in"), (12800, 1024, 256, "lstm2 in"),
                           (12800, 768, 256, "dec kvq"), (12800, 256, 256, "dec o/t"), (5120, 256, 128, "highway"),
                           (5120, 512, 128, "enc lstm in"), (5120, 224, 256, "keys1")]:
        x, W, dy = rnd(M, K), rnd(K, N) / math.sqrt(K), rnd(M, N)
        Ww = weight(W)
        out, dx, dW, db = torch.empty(M, N, device=DEV), torch.empty(M, K, device=DEV), torch.zeros(K, N, device=DEV), torch.zeros(N, device=DEV)
        f = 2.0 * M * N * K
        # algorithmic bytes: activations / gradients fp32 (read once, written once), weights from the bf16 shadow
        case("fwd  %5dx%4dx%4d %s" % (M, N, K, tag), f, lambda: ops.linear(x, Ww, None, out), lambda: ops.linear(x, W, None, out),
             4.0 * M * K + 2.0 * K * N + 4.0 * M * N)
        case("dX   %5dx%4dx%4d %s" % (M, K, N, tag), f, lambda: ops.linear_dx(dy, Ww, dx), lambda: ops.linear_dx(dy, W, dx),
             4.0 * M * N + 2.0 * K * N + 4.0 * M * K)
        ops.set_precision("bf16")
        case("dW   %5dx%4dx%4d %s (+bias)" % (K, N, M, tag), f, lambda: ops.linear_dw(x, dy, dW, db=db), None,
             4.0 * M * K + 4.0 * M * N + 2 * 4.0 * K * N)
    # ---- conv bank (16 widths, 128 -> 128 channels) and the projection convs
    B, Ti = 32, 160
    M = B * Ti
    x = rnd(M, 128)
    Ws = [rnd(k, 128, 128) / math.sqrt(k * 128) for k in range(1, 17)]
    flat = torch.cat([w.reshape(-1) for w in Ws]).contiguous()
    st = torch.zeros(flat.numel(), dtype=torch.bfloat16, device=DEV); sn = torch.zeros_like(st)
    tab, off = [], 0
    for k in range(1, 17):
        tab += [off, k, 128, 128]; off += k * 128 * 128
    ops.shadow_pack(flat, torch.tensor(tab, dtype=torch.int64, device=DEV), 16, st, sn)
    Wb = ops.Weight(flat, st, sn)
    out = torch.empty(M, 2048, device=DEV); dy = rnd(M, 2048); dx = torch.zeros(M, 128, device=DEV)
    f = 2.0 * M * 128 * 128 * 136
    nw = 136 * 128 * 128
    case("conv bank fwd 5120x2048x(136*128)", f, lambda: ops.conv_bank(x, Ti, Wb, 16, out), lambda: ops.conv_bank(x, Ti, flat, 16, out),
         4.0 * M * 128 + 2.0 * nw + 4.0 * M * 2048)
    case("conv bank dX  5120x128x(136*128)", f, lambda: ops.conv_bank_dx(dy, Ti, Wb, 16, dx), lambda: ops.conv_bank_dx(dy, Ti, flat, 16, dx),
         4.0 * M * 2048 + 2.0 * nw + 4.0 * M * 128)
    dWb = torch.zeros(flat.numel(), device=DEV)
    case("conv bank dW  17408x128x5120 (1 launch)", f, lambda: ops.conv_bank_dw(x, Ti, dy, dWb, 16), None,
         4.0 * M * 128 + 4.0 * M * 2048 + 2 * 4.0 * nw)
    mp = rnd(M, 2048); W1 = rnd(3, 2048, 128) / math.sqrt(6144); W1w = weight(W1)
    o1 = torch.empty(M, 128, device=DEV); d1 = rnd(M, 128); dmp = torch.empty(M, 2048, device=DEV); dW1 = torch.zeros(3, 2048, 128, device=DEV)
    f = 2.0 * M * 128 * 6144
    nw1 = 3 * 2048 * 128
    case("proj1 fwd 5120x128x6144 (split-K)", f, lambda: ops.conv1d(mp, Ti, W1w, o1), lambda: ops.conv1d(mp, Ti, W1, o1),
         4.0 * M * 2048 + 2.0 * nw1 + 4.0 * M * 128)
    case("proj1 dX  5120x2048x384", f, lambda: ops.conv1d_dx(d1, Ti, W1w, dmp), lambda: ops.conv1d_dx(d1, Ti, W1, dmp),
         4.0 * M * 128 + 2.0 * nw1 + 4.0 * M * 2048)
    case("proj1 dW  6144x128x5120", f, lambda: ops.conv1d_dw(mp, Ti, d1, dW1), None, 4.0 * M * 2048 + 4.0 * M * 128 + 2 * 4.0 * nw1)
    print("%-44s %7s %8s %8s %6s %7s %7s %6s %5s %6s | %9s %8s" % ("shape (M x N x K)", "GFLOP", "tile us", "TFLOP/s", "f_mfma",
                                                                  "MB", "GB/s", "f_hbm", "bound", "frac", "generic us", "TFLOP/s"))
    for r in rows:
        print("%-44s %7.2f %8.1f %8.1f %6.3f %7.1f %7.0f %6.3f %5s %6.3f | %9.1f %8.1f" % r)
    tot_f = sum(r[1] for r in rows); tot_t = sum(r[2] for r in rows); tot_b = sum(r[5] for r in rows)
    print("sum: %.1f GFLOP, %.0f MB in %.1f us = %.1f TFLOP/s (%.3f of %.0f), %.0f GB/s (%.3f of %.0f)"
          % (tot_f, tot_b, tot_t, tot_f * 1e3 / tot_t, tot_f * 1e3 / tot_t / PEAK, PEAK, tot_b * 1e3 / tot_t, tot_b * 1e3 / tot_t / HBM, HBM))
    worst = min(rows, key=lambda r: r[9]); best = max(rows, key=lambda r: r[9])
    print("roofline fraction (max of the two bounds): best %.3f (%s), worst %.3f (%s)" % (best[9], best[0].strip(), worst[9], worst[0].strip()))


if __name__ == "__main__":
    main()
