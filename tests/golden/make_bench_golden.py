"""Freezes the float64 oracle (oracle/torch_ref.py) at the EXACT benchmark workloads into compact fixtures.

BASELINE.json configs[1] (LJSpeech, B=32, Ti=160, Tm=800) and configs[3] (VCTK, B=32, Ti=80, Tm=500, 152 speakers): the
batches bench.py times (`synthetic_batch(32, 160, 800, seed=1234)`, VCTK `seed=4321`), fp32 master weights straight from
`init_params` (no slice made bf16-representable), dropout / zoneout ON with the counter-based masks of oracle/rng.py.
The full tensors are too large to commit (gradient: 25 MB, alignments: 8 MB), so each fixture keeps:

  * the three losses, the per-sample masked mel-L1 and stop-token BCE;
  * the argmax path of both alignments ([B, Td] int16) and NROW sampled (sample, step) rows of alignment1 / alignment2,
    mel (both frames of the step), stop logits and the decoder self-attention output; sampled encoder rows;
  * per parameter tensor: the gradient's L2 norm, the full gradient where the tensor has <= FULL_MAX elements, and a
    SKETCH_T-dimensional count sketch otherwise; plus one SKETCH_G-dimensional count sketch of the flat gradient.
    A count sketch is a sparse random projection (bucket h(i), sign s(i), both from a seeded numpy generator - see
    `sketch_plan`): ||S(a) - S(b)|| estimates ||a - b|| to a few percent, which is what the GPU test bounds.

The reference (TF1 + tacotron2@6af04c7) cannot run here and holds no vectors: this pins the build's own
restatement - "parity unpinned" with respect to TF stays true.  CPU only, ~15 min and ~25 GB for the LJSpeech case:
    python tests/golden/make_bench_golden.py [ljspeech] [vctk] [ljspeech_sharp_lo] [ljspeech_sharp_hi]
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NROW = 64
FULL_MAX = 2048
SKETCH_T = 1024
SKETCH_G = 4096
PARAM_SEED = 3
RNG_SEED = 5

# The converged regime (r6): training drives alignment 1 towards one-hot rows that ride on the `+1e-7` floor of the forward recursion
# (reference modules/forward_attention.py:107-121) with large energies, while everything init_params produces is diffuse (mean row
# entropy 2.95 nats of log 160 = 5.08, max alpha 0.51).  `sharpen` reshapes the location-sensitive score of the SAME random model so
# that its rows are sharp, on both sides of the kernels' softmax-form switch at sum|v| = 30 (csrc/attn_cluster.hip: constant shift /
# lazy normalisation below it, the in-chain maximum above):
#   sharp_lo: v concentrated on its first 4 units and scaled to sum|v| = 29.5, query layer and keys x 8 (saturated tanh: neighbouring
#             energies differ by several units) - mean entropy ~0.4 nats, max alpha > 0.95 on ~45 % of the rows (what sum|v| < 30 allows
#             a random network);
#   sharp_hi: v x 24 (sum|v| = 443), query layer and keys x 4 - mean entropy ~0.2 nats, max alpha > 0.95 on ~70 % of the rows.
CASES = {
    "ljspeech": dict(cfg=dict(), batch=dict(B=32, Ti=160, Tm=800, seed=1234)),
    "ljspeech_sharp_lo": dict(cfg=dict(), batch=dict(B=32, Ti=160, Tm=800, seed=1234), sharpen=dict(keep=4, sv=1.6, sq=8.0, sk=8.0)),
    "ljspeech_sharp_hi": dict(cfg=dict(), batch=dict(B=32, Ti=160, Tm=800, seed=1234), sharpen=dict(keep=0, sv=24.0, sq=4.0, sk=4.0)),
    "vctk": dict(cfg=dict(num_speakers=152, speaker_offset=225),
                 batch=dict(B=32, Ti=80, Tm=500, seed=4321, min_source_length=30, min_target_steps=90, num_speakers=152,
                            speaker_offset=225)),
}


def sample_rows(B, Td, seed):
    """NROW (sample, decoder step) pairs: the first / last steps of a few samples plus uniform draws"""
    g = np.random.default_rng(seed)
    b = g.integers(0, B, NROW)
    t = g.integers(0, Td, NROW)
    t[:4] = 0
    t[4:8] = Td - 1
    t[8:12] = 1
    return b.astype(np.int32), t.astype(np.int32)


def crc_of(arrays):
    """CRC-32 over the bytes of a dict of arrays in key order (pins the seeded inputs the vectors belong to)"""
    import zlib
    c = 0
    for k in sorted(arrays):
        c = zlib.crc32(np.ascontiguousarray(arrays[k]).tobytes(), zlib.crc32(k.encode(), c))
    return c


def make_batch(kw):
    from satt_amd.datasets.synthetic import synthetic_batch
    kw = dict(kw)
    return synthetic_batch(kw.pop("B"), kw.pop("Ti"), kw.pop("Tm"), **kw)


def sharpen_params(P, keep=0, sv=1.0, sq=1.0, sk=1.0):
    """the `sharpen` transformation of a parameter dict (a copy): dec.att1.v concentrated on its first `keep` units (0: all) and scaled
    to sv x its initial sum of magnitudes, the fused query layer dec.att.Wq x sq, the key layer dec.att1.Wm x sk"""
    P = dict(P)
    v0 = np.asarray(P["dec.att1.v"], dtype=np.float64)
    v = v0.copy()
    if keep > 0:
        v[int(keep):] = 0.0
    v *= sv * np.abs(v0).sum() / np.abs(v).sum()
    P["dec.att1.v"] = v.astype(np.float32)
    P["dec.att.Wq"] = (np.asarray(P["dec.att.Wq"], dtype=np.float64) * sq).astype(np.float32)
    P["dec.att1.Wm"] = (np.asarray(P["dec.att1.Wm"], dtype=np.float64) * sk).astype(np.float32)
    return P


def build(name):
    import satt_amd  # noqa: F401
    from satt_amd.params import ModelConfig, init_params
    from common import count_sketch, oracle_run
    case = CASES[name]
    cfg = ModelConfig(**case["cfg"])
    P = init_params(cfg, PARAM_SEED)
    if "sharpen" in case:
        P = sharpen_params(P, **case["sharpen"])
    batch = make_batch(case["batch"])
    t0 = time.time()
    out, col, g = oracle_run(case["cfg"], P, batch, True, seed=RNG_SEED)
    print("%s: oracle forward + backward %.0f s, loss %.9f" % (name, time.time() - t0, float(out["loss"])), flush=True)
    B, Td = batch["done"].shape
    r = cfg.r
    mel = out["mel"].detach().numpy()
    stop = out["stop"].detach().numpy().reshape(B, Td)
    al1, al2 = out["alignment1"].detach().numpy(), out["alignment2"].detach().numpy()
    tgt = np.asarray(batch["mel"], dtype=np.float64)
    w = np.asarray(batch["spec_loss_mask"], dtype=np.float64)
    per_mel = (np.abs(mel - tgt).mean(-1) * w).sum(-1) / np.maximum(w.sum(-1), 1.0)
    sb, st = sample_rows(B, Td, 99)
    eb = np.random.default_rng(98).integers(0, B, NROW).astype(np.int32)
    et = np.array([np.random.default_rng(97 + i).integers(0, int(batch["source_length"][b])) for i, b in enumerate(eb)], dtype=np.int32)
    keep = dict(loss=np.float64(out["loss"].detach()), mel_loss=np.float64(out["mel_loss"].detach()),
                done_loss=np.float64(out["done_loss"].detach()), per_sample_mel_l1=per_mel,
                path1=al1.argmax(-1).astype(np.int16), path2=al2.argmax(-1).astype(np.int16),
                rows_b=sb, rows_t=st, align1_rows=al1[sb, st].astype(np.float32), align2_rows=al2[sb, st].astype(np.float32),
                mel_rows=mel.reshape(B, Td, r * cfg.num_mels)[sb, st].astype(np.float32), stop_rows=stop[sb, st].astype(np.float32),
                dec_out_rows=col["dec_out"].detach().numpy()[sb, st].astype(np.float32),
                enc_b=eb, enc_t=et, lstm_out_rows=out["lstm_out"].detach().numpy()[eb, et].astype(np.float32),
                sa_out_rows=out["sa_out"].detach().numpy()[eb, et].astype(np.float32),
                align1_mean_entropy=np.float64(-(al1 * np.log(np.maximum(al1, 1e-300))).sum(-1).mean()),
                align1_max_mean=np.float64(al1.max(-1).mean()), align1_frac_max_above_095=np.float64((al1.max(-1) > 0.95).mean()),
                att1_v_abs_sum=np.float64(np.abs(np.asarray(P["dec.att1.v"], dtype=np.float64)).sum()))
    print("%s: alignment-1 mean row entropy %.3f nats, mean max %.3f, rows with max > 0.95: %.3f, sum|v| = %.1f"
          % (name, keep["align1_mean_entropy"], keep["align1_max_mean"], keep["align1_frac_max_above_095"], keep["att1_v_abs_sum"]), flush=True)
    names = list(g.keys())
    flat = np.concatenate([np.asarray(g[k], dtype=np.float64).ravel() for k in names])
    keep["grad_names"] = np.array(names)
    keep["grad_norms"] = np.array([np.linalg.norm(np.asarray(g[k], dtype=np.float64)) for k in names])
    keep["grad_norm_all"] = np.float64(np.linalg.norm(flat))
    keep["grad_sketch_all"] = count_sketch(flat, SKETCH_G, 0)
    for i, k in enumerate(names):
        a = np.asarray(g[k], dtype=np.float64)
        if a.size <= FULL_MAX:
            keep["grad_full." + k] = a.astype(np.float32)
        else:
            keep["grad_sketch." + k] = count_sketch(a, SKETCH_T, i + 1).astype(np.float32)
    meta = dict(param_seed=PARAM_SEED, rng_seed=RNG_SEED, nrow=NROW, full_max=FULL_MAX, sketch_t=SKETCH_T, sketch_g=SKETCH_G,
                batch_crc=crc_of(batch), param_crc=crc_of(P))
    path = os.path.join(HERE, "bench_%s.npz" % name)
    np.savez_compressed(path, **{"meta." + k: np.int64(v) for k, v in meta.items()}, **keep)
    print("%s: wrote %s (%.0f KB)" % (name, path, os.path.getsize(path) / 1024), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("SATT_ORACLE_THREADS", "6")))
    for n in (sys.argv[1:] or list(CASES)):
        build(n)
