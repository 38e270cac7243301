"""Per-phase wall-clock breakdown of the persistent attention-RNN kernels (profile build of the library)."""
import ctypes, os, subprocess, sys
sys.path.insert(0, '.')
ROOT = os.getcwd()
src = os.path.join(ROOT, "self-attention-tacotron_amd", "csrc")
out = os.environ.get("SATT_PROF_LIB", "/tmp/libsatt_prof.so")     # SATT_PROF_LIB: a profile build made beforehand (no hipcc run here)
if not os.environ.get("SATT_PROF_LIB"):
  subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSATT_PROFILE"] + (["-DSATT_TRACE_BWD"] if os.environ.get("SATT_TRACE_BWD") else []) + (["-DSATT_TRACE_ONLY"] if os.environ.get("SATT_TRACE_ONLY") else []) +
                      [os.path.join(src, f) for f in ("gemm.hip", "gemm_tile.hip", "flash.hip", "small_attn.hip", "elementwise.hip", "lstm.hip", "lstm_cluster.hip", "attn_rnn.hip", "attn_cluster.hip", "decode.hip", "api.hip")] + ["-o", out])
import torch
import satt_amd
from satt_amd import _lib
_lib.LIB_PATH = out
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch
import os
eng = Engine(ModelConfig(), "cuda", param_seed=0, rng_seed=1)
if os.environ.get("SATT_CMAX"):
    ops.ATTN_CLUSTER_SIZES = tuple(int(x) for x in os.environ["SATT_CMAX"].split(","))
if os.environ.get("SATT_CHUNKS"):
    eng.pipeline_chunks = int(os.environ["SATT_CHUNKS"])
b = eng.to_device_batch(synthetic_batch(32, 160, 800, seed=1234))
for _ in range(2):
    ctx = eng.train_step(b)
torch.cuda.synchronize()
l = _lib.lib()
buf = (ctypes.c_ulonglong * 32)()
rd = l.satt_prof_read_cluster if (eng.use_clusters and ctx["att_cluster"][0]) else l.satt_prof_read
rd.argtypes = [ctypes.c_void_p]
rd(buf)
print("cluster size:", ctx["att_cluster"][0])
v = list(buf)
print("len(b=0) =", int(b["source_length"][0]))
names_f = ["loop-top/xg", "MFMA Wrec", "cell + partial-pq MFMA", "loc-conv", "X1 gather", "energies", "local softmax", "ctx MFMA + X2", "normalise (lazy: reciprocals)"]
names_b = ["loop-top", "(a) load state", "(b) dalpha + conv bwd + Xb", "(c) softmax bwd", "(d) energy bwd + Xd", "dpq reduce", "(f) dq MFMA", "(g) cell bwd", "(h) dvec MFMA + Xh"]
print("FWD per step (us):")
for n, x in zip(names_f, v[:9]): print("  %-28s %7.2f" % (n, x / 100.0 / 400))
print("  [energies above = the tail after the rows; in front of it: setup %.2f, rows %.2f]" % (v[9] / 100.0 / 400, v[10] / 100.0 / 400))
print("  total %.2f" % ((sum(v[:9]) + v[9] + v[10]) / 100.0 / 400))
print("BWD per step (us):")
for n, x in zip(names_b, v[16:25]): print("  %-28s %7.2f" % (n, x / 100.0 / 400))
print("  total %.2f" % (sum(v[16:32]) / 100.0 / 400))
print("  extra marks (slots 9..15, us per step; each is the time since the previous mark of the step):", [round(x / 100.0 / 400, 2) for x in v[25:32]])

# ---- exchange trace of the forward kernel (members of sample 0): skew vs mechanism
if os.environ.get("SATT_TRACE"):
    import numpy as np
    n = 8 * 128 * 16
    tb = (ctypes.c_ulonglong * n)()
    l.satt_prof_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    l.satt_prof_read_trace(tb, n)
    TT = np.array(list(tb), dtype=np.float64).reshape(8, 128, 16)[:4, 8:120, :8] / 100.0
    if TT[:, :, 5].max() > 0:       # (slots 5..7 are written by the in-chain normalisation only: the lazy forward of r5 has none)
        print("normalise: got2->scalars %s  scalars->rows %s  rows->ctx %s  ctx->barrier-done %s" % tuple(
            np.round(x.mean(1), 2) for x in (TT[:, :, 5] - TT[:, :, 3], TT[:, :, 6] - TT[:, :, 5], TT[:, :, 7] - TT[:, :, 6], TT[:, :, 4] - TT[:, :, 7])))
    else:
        print("normalise: lazy form (r5) - behind the exchange X2 only the three reciprocals of the sums are left on the chain")
    T = TT[:, :, :5]
    pub1, got1, pub2, got2, end = (T[:, :, i] for i in range(5))
    print("X1: last publish - own publish (skew) per member:", np.round((pub1.max(0)[None] - pub1).mean(1), 2))
    print("X1: gather done - last publish (mechanism)       :", np.round((got1 - pub1.max(0)[None]).mean(1), 2))
    print("X2: skew per member                              :", np.round((pub2.max(0)[None] - pub2).mean(1), 2))
    print("X2: gather done - last publish                   :", np.round((got2 - pub2.max(0)[None]).mean(1), 2))
    print("step time (end to end)                           :", np.round(np.diff(end, axis=1).mean(1), 2))
    print("end(prev) -> publish1 (MFMA, cell, partial pq)   :", np.round((pub1[:, 1:] - end[:, :-1]).mean(1), 2))
    print("got1 -> publish2 (energies, softmax, ctx MFMA)   :", np.round((pub2 - got1).mean(1), 2))
    print("got2 -> end (normalise)                          :", np.round((end - got2).mean(1), 2))
    print("publish1 -> got1 (own view)                      :", np.round((got1 - pub1).mean(1), 2))
    print("publish2 -> got2 (own view)                      :", np.round((got2 - pub2).mean(1), 2))

if os.environ.get("SATT_TRACE_BWD"):
    import numpy as np
    n = 8 * 128 * 16
    tb = (ctypes.c_ulonglong * n)()
    l.satt_prof_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    l.satt_prof_read_trace(tb, n)
    T = np.array(list(tb), dtype=np.float64).reshape(8, 128, 16)[:4, 8:120, :12] / 100.0
    names = ["(a) load+barrier", "(b) compute->publish", "(e) conv + Xb gather+barrier", "(c) softmax bwd", "(d) rows->publish", "Xd gather+barrier",
             "dpq sum", "(f) dq MFMA", "(g) cell", "(h) MFMA+barrier", "(h) reduce+publish", "Xh gather+barrier+dh"]
    prev = np.concatenate([T[:, :1, 11] * np.nan, T[:, :-1, 11]], axis=1)
    seg = [T[:, :, 0] - prev] + [T[:, :, i] - T[:, :, i - 1] for i in range(1, 12)]
    print("BWD segments (us), members 0..3:")
    for nme, sg in zip(names, seg):
        print("  %-24s %s" % (nme, np.round(np.nanmean(sg, axis=1), 2)))
    print("  step:", np.round(np.diff(T[:, :, 11], axis=1).mean(1), 2))
    for nm_, pub, got in (("Xb", 1, 2), ("Xd", 4, 5), ("Xh", 10, 11)):
        lastpub = T[:, :, pub].max(0)[None]
        print("  %s: skew %s   gather-done - last publish %s" % (nm_, np.round((lastpub - T[:, :, pub]).mean(1), 2), np.round((T[:, :, got] - lastpub).mean(1), 2)))

if os.environ.get("SATT_PROLOG"):
    pb = (ctypes.c_ulonglong * 8)()
    l.satt_prof_read_prolog.argtypes = [ctypes.c_void_p]
    l.satt_prof_read_prolog(pb)
    pv = [x / 100.0 for x in pb]
    print("FWD prologue (us) [weights->regs/LDS, wq+tables+zero, keys+values staging, handshake incl. barrier, restart]:", [round(pv[i + 1] - pv[i], 1) for i in range(5)])
