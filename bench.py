#!/usr/bin/env python
"""bench.py — teacher-forced Self-attention Tacotron train step on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W          (ONE command for any N: with WORLD_SIZE unset and N > 1 this process
                                                          re-executes itself under torch.distributed.run, one rank per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (the driver's own launch line: taken as it is)

A "step" = forward + masked-L1/BCE loss + backward + global-norm clip + TF-Adam (+ gradient all-reduce over RCCL,
three buckets in the order the backward pass finishes them, overlapped with it: parallel.py) on one synthetic
LJSpeech-shaped batch of 32 utterances per GPU (BASELINE.json configs[1]; weak scaling).  Prints ONE JSON line on rank 0.
The run FAILS (non-zero exit, no JSON line) if the process group's world size is not N or fewer than N devices are visible:
`n_gpus` is what the process group says, never the flag.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_TBS = 8.0           # HBM3E peak (spec; /opt/skills/guides/MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0       # dense MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def train_flops(B, Ti, Td):
    """Algorithmic FLOPs of one train step: 3 x forward; forward MACs per unit from §8d."""
    enc = 3637248 + 2 * (Ti - 160) * 32                                             # per encoder position
    loop = 73728 + 688128 + 57344 + Ti * 1650 + 8192 + Ti * 64 + 819200 + 524288      # per (sample, step)
    post = 4 * 256 * 256 + 2 * Td * 256 + 256 * 256 + 256 * 161                          # per (sample, step)
    fwd = 2.0 * B * (Ti * enc + Td * (loop + post))
    return 3.0 * fwd


def attn_loop_flops(B, Ti, Td, backward):
    """Algorithmic FLOPs executed INSIDE the persistent attention-RNN kernel per launch:
    recurrent gate mat-vec (544x1024), query projections (256x256), location conv + energies (Ti*(50+1120+224)),
    additive energies (Ti*32), contexts (Ti*(256+32)); backward ~ 2x (transposed mat-vecs + recompute)."""
    per_step = 544 * 1024 + 256 * 256 + Ti * (50 + 1120 + 224 + 32 + 288)
    f = 2.0 * B * Td * per_step
    return 2.0 * f if backward else f


def _cpu_baseline_worker():
    """(child process) the PyTorch-CPU restatement (oracle/torch_ref.py) timed as BASELINE.md §3 asks: a FULL train step =
    forward + masked-L1/BCE loss + backward + global-norm clip + TF-Adam (dropout / zoneout on, the rate schedule), one
    untimed warm-up step, then the median of the timed steps; BASELINE config 1 (B=8) and the GPU run's own batch (B=32)."""
    import torch
    from oracle import torch_ref
    import satt_amd  # noqa: F401
    from satt_amd.params import ModelConfig, init_params
    from satt_amd.datasets.synthetic import synthetic_batch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cfg = torch_ref.Cfg()
    Ti, Tm = 160, 800
    P0 = init_params(ModelConfig(), 0)

    def make(B):
        Pt = torch_ref.to_torch(P0, torch.float32, requires_grad=True)
        m = {k: torch.zeros_like(v) for k, v in Pt.items()}
        v = {k: torch.zeros_like(x) for k, x in Pt.items()}
        host = synthetic_batch(B, Ti, Tm, seed=1234)
        return Pt, m, v, torch_ref.batch_to_torch(host, torch.float32), int(host["target_length"].sum())

    def step(state, t):
        Pt, m, v, bt, _ = state
        out = torch_ref.forward(Pt, bt, cfg, True, t)
        grads = torch.autograd.grad(out["loss"], list(Pt.values()), allow_unused=True)
        g = {k: (gr if gr is not None else torch.zeros_like(p)) for (k, p), gr in zip(Pt.items(), grads)}
        with torch.no_grad():
            torch_ref.clip_and_adam(Pt, g, m, v, t, torch_ref.learning_rate(5e-4, t - 1))

    def timed(state, n, t0=2):
        ts = []
        for i in range(n):
            a = time.time(); step(state, t0 + i); ts.append(time.time() - a)
        return ts
    # thread count: the graph is thousands of tiny per-step ops, so more threads mostly add synchronisation; one step each
    # at a few counts (this doubles as the warm-up), keep the fastest
    st8 = make(8)
    step(st8, 1)                                    # untimed: first-touch / allocator warm-up
    sweep = {}
    for n in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(n)
        a = time.time(); step(st8, 1); sweep[n] = time.time() - a
    ncores = min(sweep, key=sweep.get)
    torch.set_num_threads(ncores)
    t8 = timed(st8, 5)
    st32 = make(32)
    step(st32, 1)                                   # warm-up
    t32 = timed(st32, 3)
    med = lambda x: sorted(x)[len(x) // 2]
    rows = []
    for B, ts, st in ((8, t8, st8), (32, t32, st32)):
        rows.append({"batch": B, "s_per_step_median": med(ts), "padded_mel_frames_per_sec": B * Tm / med(ts),
                     "valid_mel_frames_per_sec": st[4] / med(ts), "timed_steps": len(ts)})
    print(json.dumps({"value": rows[1]["padded_mel_frames_per_sec"], "unit": "mel-frames/sec", "cores": ncores, "kind": "port",
                      "rows": rows, "thread_sweep_s_per_step_b8": {str(k): round(v, 3) for k, v in sweep.items()},
                      "sample": "full train steps (fwd + loss + bwd + clip + TF-Adam, dropout / zoneout on) of the fp32 "
                                "PyTorch-CPU restatement oracle/torch_ref.py on the synthetic Ti=%d, Tm=%d batch: B=8 (BASELINE "
                                "config 1): 1 warm-up + 5 timed, median %.2f s; B=32 (the GPU run's batch; `value`): 1 warm-up + 3 "
                                "timed, median %.2f s; %d threads (fastest of the sweep; %d CPUs visible)"
                                % (Ti, Tm, med(t8), med(t32), ncores, avail)}))


def cpu_baseline(timeout_s=400):
    """The oracle's PyTorch-CPU restatement (kind 'port') timed on this host in a child process with a hard
    timeout, so the default bench run always finishes within minutes."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], capture_output=True,
                           text=True, timeout=timeout_s, cwd=ROOT)
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "unit": "mel-frames/sec", "cores": 0, "kind": "port",
                "sample": "worker failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "mel-frames/sec", "cores": 0, "kind": "port",
                "sample": "oracle step did not finish within %d s on this host" % timeout_s}


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that has one (committed summaries of rocprofv3 runs), or None"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return fs[-1] if fs else None


def _log(msg):
    print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def decode_bench(eng, steps=200, Ti=100):
    """BASELINE config 5: free-running autoregressive decode, B=1, hipGraph-captured step (inference.DecodeSession);
    ms per decoder step from HIP events around the replay loop (encoder and result copies excluded)."""
    import numpy as np
    from satt_amd.inference import infer
    g = np.random.default_rng(1234)
    src = g.integers(1, 68, (1, Ti)); src[:, 0] = 0; src[:, -1] = 0
    sl = np.full((1,), Ti, dtype=np.int64)
    kw = dict(max_steps=steps, min_steps=10 ** 6)
    infer(eng, src, sl, **kw)                 # builds the session (buffers + captured graph)
    ms = sorted(infer(eng, src, sl, **kw)["decode_ms"] for _ in range(3))[1]
    r = eng.cfg.r
    ses = next(reversed(eng._decode_sessions.values()))
    how = ("persistent step kernel, launches of up to %d steps" % ses.K) if ses.mega is not None else "hipGraph of 8 steps per replay"
    # what ms_per_step covers: the decoder steps alone (HIP events around the launch loop of inference.infer).  The per-utterance
    # prologue is OUTSIDE it: encoder forward, memory keys, the folded output transform and - persistent kernel - the four
    # [Ti, 1024] context-table GEMMs that r5 moved out of the step (values W_c per utterance): `utterance_ms` has everything
    t0 = time.perf_counter()
    infer(eng, src, sl, **kw)
    import torch
    torch.cuda.synchronize()
    utt_ms = 1e3 * (time.perf_counter() - t0)
    return {"workload": "free-running decode, B=1, Ti=%d, %d decoder steps, %s; ms_per_step = the decoder steps alone (HIP events), "
                        "EXCLUDING the per-utterance prologue (encoder, memory keys, folded output transform, context-table GEMMs): "
                        "utterance_ms is the whole call incl. result copies" % (Ti, steps, how),
            "utterance_ms": utt_ms,
            "ms_per_step": ms / steps, "mel_frames_per_sec": steps * r / (ms * 1e-3),
            "realtime_factor": (ms * 1e-3) / (steps * r * 0.0125),
            "launches_per_step": (1.0 / ses.K) if ses.mega is not None else max(x.kernel_launches for x in eng._decode_sessions.values())}


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _spawn_ranks(n, argv):
    """`bench.py --gpus N` without a launcher (reference train.py:16,68: --multi-gpus is ONE command): re-execute this
    script under torch.distributed.run, one rank per GPU of this node, rendezvous on 127.0.0.1; the exit code is the job's."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes of one node needs it here
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def _dry_run(args, dp, world, rank):
    """(test hook, --dry-run) everything of the N-rank run but the engine: rendezvous, world-size check, the three-bucket
    exchange of a small flat buffer, barrier + max-over-ranks timing, the JSON line - runs on CPU over gloo (tests/test_dp_gloo.py)"""
    import torch
    import torch.distributed as dist
    n = 3000
    g = torch.zeros(n)
    dp.bind(g)
    dp.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.copy_(torch.arange(n, dtype=torch.float32) % 32 * (rank + 1))
        dp.allreduce(2000, n); dp.allreduce(1000, 2000); dp.allreduce(0, 1000)
        dp.wait()
    dp.barrier()
    dt = time.perf_counter() - t0
    ok = bool(torch.equal(g, torch.arange(n, dtype=torch.float32) % 32 * sum(r + 1 for r in range(world))))
    per_rank_ms = [1e3 * x / args.steps for x in dp.gather_over_ranks(dt)]
    dt = dp.max_over_ranks(dt)
    ranks = [None] * world
    if dp.active:
        dist.all_gather_object(ranks, {"rank": rank, "pid": os.getpid(), "backend": dist.get_backend()})
    else:
        ranks = [{"rank": 0, "pid": os.getpid(), "backend": None}]
    if rank == 0:
        print(json.dumps({"metric": "dry run (no engine): three-bucket exchange of a 3000-float buffer", "dry_run": True, "value": None,
                          "n_gpus": dist.get_world_size() if dp.active else 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * dt / args.steps, "per_rank_ms": [round(x, 4) for x in per_rank_ms],
                          "rccl_ranks": ranks, "sum_ok": ok}))
    dp.shutdown()
    if not ok:
        raise SystemExit("dry run: the bucketed all-reduce did not sum over the ranks")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the free-running decode measurement (BASELINE config 5)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--chunks", type=int, default=None, help="time chunks of the recurrent stream pipeline")
    ap.add_argument("--time-all-kernels", action="store_true",
                    help="HIP-event brackets around all eight recurrent kernel families instead of the two attention "
                         "kernels only (70 instead of 4 events per step: costs ~0.1 ms per step)")
    ap.add_argument("--chunked-attention", action="store_true",
                    help="one attention launch per pipeline chunk instead of one per direction (for counter-"
                         "collecting profiler passes, which serialise kernels)")
    ap.add_argument("--model", default="self-attention-tacotron", choices=["self-attention-tacotron", "tacotron", "vctk"],
                    help="tacotron = the baseline ExtendedTacotronV1Model (examples/ljspeech/tacotron.json); vctk = BASELINE "
                         "config 4 (examples/vctk/self-attention-tacotron.json: 152 speakers, multi-speaker decoder pre-net); "
                         "the headline metric is the default")
    ap.add_argument("--tail", default=None, help=argparse.SUPPRESS)       # tuning: Engine.pipeline_tail as "n,div"
    ap.add_argument("--wgrad-defer", type=int, default=None, help=argparse.SUPPRESS)     # tuning: Engine.wgrad_defer
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", default=None, help=argparse.SUPPRESS)        # test hook: "gloo" exchanges device tensors
    ap.add_argument("--share-device", action="store_true", help=argparse.SUPPRESS)   # test hook: every rank on cuda:0
    ap.add_argument("--force-dist", action="store_true", help=argparse.SUPPRESS)     # diagnostics: a one-rank RCCL group at N=1
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)        # test hook: the N-rank plumbing without the engine
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return _cpu_baseline_worker()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one command for N GPUs: this process becomes the launcher of its N ranks
        if not (args.dry_run or args.share_device):
            import torch
            if torch.cuda.device_count() < args.gpus:
                raise SystemExit("--gpus %d but only %d device(s) visible" % (args.gpus, torch.cuda.device_count()))
        raise SystemExit(_spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import satt_amd  # noqa: F401
    from satt_amd import ops
    from satt_amd.engine import Engine
    from satt_amd.params import ModelConfig
    from satt_amd.datasets.synthetic import synthetic_batch
    from satt_amd.parallel import DataParallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.share_device:
        local = 0
    if args.dry_run:
        dp = DataParallel(world, rank, local, backend=args.backend or "gloo")
    else:
        if not args.share_device and torch.cuda.device_count() < world:
            raise SystemExit("--gpus %d but only %d device(s) visible to rank %d" % (args.gpus, torch.cuda.device_count(), rank))
        torch.cuda.set_device(local)
        dp = DataParallel(world, rank, local, backend=args.backend, force=args.force_dist)
    if dp.active:
        import torch.distributed as dist
        if dist.get_world_size() != args.gpus and not (args.force_dist and args.gpus == 1):
            raise SystemExit("--gpus %d but the process group has %d rank(s)" % (args.gpus, dist.get_world_size()))
    if args.dry_run:
        return _dry_run(args, dp, world, rank)
    n_gpus = dist.get_world_size() if dp.active else 1
    # who the ranks are: (rank, device index, device name, PCI bus id) of every member of the group, as rank 0 prints them
    prop = torch.cuda.get_device_properties(local)
    me = {"rank": rank, "device": local, "name": prop.name, "pci_bus_id": getattr(prop, "pci_bus_id", None), "pid": os.getpid()}
    rccl_ranks = [me]
    if dp.active:
        rccl_ranks = [None] * dist.get_world_size()
        dist.all_gather_object(rccl_ranks, me)
    ops.set_precision(args.precision)

    # BASELINE config 4 (VCTK) has its own shape (Ti <= 80, Tm <= 500); everything else is the LJSpeech shape
    B, Ti, Tm = (args.batch, 80, 500) if args.model == "vctk" else (args.batch, 160, 800)
    cfg = {"self-attention-tacotron": lambda: ModelConfig(),
           "tacotron": lambda: ModelConfig(sa_units=0, att2_units=0, dec_sa_units=0, att1_units=256),
           "vctk": lambda: ModelConfig(num_speakers=152, speaker_offset=225)}[args.model]()
    eng = Engine(cfg, "cuda:%d" % local, param_seed=0, rng_seed=1234)
    dp.bind(eng.grad)
    if args.chunks:
        eng.pipeline_chunks = args.chunks
    if args.tail:
        eng.pipeline_tail = tuple(int(v) for v in args.tail.split(","))
    if args.wgrad_defer is not None:
        eng.wgrad_defer = bool(args.wgrad_defer)
    if args.chunked_attention:
        eng.single_launch_attention = False
    host_batch = synthetic_batch(B, Ti, Tm, seed=1234 + rank, **(dict(min_source_length=30, min_target_steps=90) if args.model == "vctk" else {}))
    if cfg.num_speakers > 0:
        import numpy as np
        host_batch["speaker_id"] = (np.random.default_rng(rank).integers(0, cfg.num_speakers, B) + cfg.speaker_offset).astype(np.int64)
    batch = eng.to_device_batch(host_batch)
    Td = Tm // cfg.r

    def step():
        ctx = eng.train_step(batch, allreduce=dp.allreduce if dp.active else None)
        dp.wait()
        eng.optimizer_step(grad_scale=1.0 / world)
        return ctx

    # The per-kernel HIP-event timing below keeps ~70 timing events per step alive until the summary; the runtime
    # grows its pool of profiling signals in one ~40 ms host stall when that happens inside the timed steps (measured:
    # one 23 ms step among 11.6 ms ones).  Grow the pool here, before the warm-up, instead: instrumentation only.
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(80 * args.steps + 256)]
    for e in pool:
        e.record()
    torch.cuda.synchronize()
    del pool
    _log("engine ready (%d params), warmup" % eng.nparam)
    for i in range(args.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize(); _log("first step done")
    eng.timing = {}
    if dp.active:
        dp.timing = []
    if not args.time_all_kernels:     # the dominant kernel only (backward attention loop)
        eng.timing_names = {"attn_rnn_bwd"}       # (every bracket is two marker packets on the launching stream, ~5 us each)
    dp.barrier(); torch.cuda.synchronize()
    skipped0 = [int(float(v)) for v in eng.opt_state[-2:].tolist()]     # sticky device counters: updates skipped / for a non-finite gradient
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    marks[0].record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    torch.cuda.synchronize(); dp.barrier()
    dt = time.perf_counter() - t0
    per = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    _log("per-step ms: " + " ".join("%.2f" % x for x in per))
    per_rank_ms = [1e3 * x / args.steps for x in dp.gather_over_ranks(dt)]
    dt = dp.max_over_ranks(dt)
    _log("timed %d steps: %.2f ms/step" % (args.steps, 1e3 * dt / args.steps))
    timing = eng.timing_summary()
    eng.timing = None
    ar_ms, ar_wait_ms = dp.timing_summary() if dp.active else (0.0, 0.0)
    dp.timing = None
    loss = float(eng.losses[2])
    # an update the device skipped (a cluster kernel's exchange time-out poisons the step's gradient; a non-finite gradient) would make
    # a timed step cheaper than a real one: counted on the device, reported, and the line says so when it is not zero
    skipped = [int(float(v)) - v0 for v, v0 in zip(eng.opt_state[-2:].tolist(), skipped0)]

    if rank == 0:
        ms = 1e3 * dt / args.steps
        frames = world * B * Tm                      # padded mel frames per step, whole job
        valid = int(batch["target_length"].sum()) * world
        per_step = {k: v[0] / args.steps for k, v in timing.items()}          # ms per train step (all launches)
        dom = max(per_step, key=per_step.get) if per_step else None
        roof = None
        if dom is not None:
            dms = per_step[dom]
            if dom.startswith("attn_rnn"):
                fl = attn_loop_flops(B, Ti, Td, dom.endswith("bwd"))
            else:
                H = 256 if dom.startswith("lstm") else 128
                nd = 1 if dom.startswith("lstm") else 2
                T_ = Td if dom.startswith("lstm") else Ti
                fl = 2.0 * B * T_ * nd * H * 4 * H * (1.0 if dom.endswith("fwd") else 1.0)
            ach = fl / (dms * 1e-3) / 1e12
            nl = max(1, timing[dom][1] // args.steps)
            traffic, tsrc = None, None
            # HBM bytes per launch from the committed rocprofv3 PMC passes of this round (tools/make_pmc_traffic.py: separate
            # FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH correction); null if the file is not there
            tfile = _latest_profile("pmc_traffic.json")
            if tfile:
                pmc = json.load(open(tfile))
                if dom in pmc:
                    traffic = pmc[dom]["hbm_bytes_per_step"] / nl if "hbm_bytes_per_step" in pmc[dom] \
                        else pmc[dom]["hbm_bytes_per_launch"]
                    # counter collection serialises kernels: the PMC passes run `bench.py --chunks 1` - ONE attention launch per
                    # direction over all steps behind the complete LSTM launches, i.e. the timed kernel without its chunk waits
                    tsrc = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --chunks 1`: " \
                           "%d launch(es) per step there, %d in the timed run; bytes per step / launches of the timed run)" \
                           % (os.path.basename(tfile), round(pmc[dom].get("launches_per_step_profiled", 1)), nl)
            hbm_frac = (traffic / (dms / nl * 1e-3) / 1e12 / PEAK_HBM_TBS) if traffic else None
            # "bound": neither roof binds this kernel - it is a serial chain of 400 dependent steps (say so).  frac
            # stays the fraction of the dense bf16 MFMA peak its algorithmic FLOPs reach (the contract's field); hbm_frac is the
            # counter traffic against 8 TB/s
            roof = {"kernel": dom, "bound": "latency", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "hbm_frac": hbm_frac, "traffic_source": tsrc,
                    "flops_per_launch": fl / nl, "launch_ms": dms / nl, "ms_per_step": dms, "launches_per_step": nl,
                    "note": "latency-bound persistent recurrence (cluster of 4 workgroups per sample, %d CUs, weights "
                            "register-resident): achieved = algorithmic FLOPs of a launch / its HIP-event duration "
                            "(measured live on the launching stream); neither MFMA nor HBM is the limiter - the "
                            "serial step chain (2 cross-workgroup exchanges + ~8 workgroup barriers per step, ~50 %% of the "
                            "wave cycles parked) is; see DESIGN.md 3 and profiles/r06_attn_loop_phases.txt" % (4 * B)}
        step_tflops = train_flops(B, Ti, Td) * world / (ms * 1e-3) / 1e12
        line = {
            "metric": "mel-frames/sec (teacher-forced train step)", "value": frames / (dt / args.steps),
            "unit": "mel-frames/sec", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": ("VCTK self-attention-tacotron" if args.model == "vctk" else "LJSpeech %s" % args.model) + ".json, teacher-forced train step "
                                   "(fwd+loss+bwd+clip+Adam), B=%d/GPU, Ti=%d, Tm=%d, r=2" % (B, Ti, Tm),
                       "global_batch": world * B, "parallelism": "dp%d" % world},
            "ms_per_step_median": sorted(per)[len(per) // 2],
            "valid_mel_frames_per_sec": valid / (dt / args.steps),
            "step_tflops": step_tflops, "step_frac_of_bf16_peak": step_tflops / (PEAK_BF16_TFLOPS * world),
            "loss": loss,
            "skipped_updates": {"timed_steps": args.steps, "skipped": skipped[0], "non_finite": skipped[1]},
            "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step.items())},
            # N > 1: time inside the two gradient-bucket all-reduces (HIP events on their issuing streams; includes waiting for
            # the slowest peer) and the part of it the optimiser's stream actually stalled for (events around its wait)
            "allreduce_ms": (ar_ms / args.steps) if dp.active else None,
            "exposed_allreduce_ms": (ar_wait_ms / args.steps) if dp.active else None,
            # wall time per step of EVERY rank (value uses the slowest), the bucket plan and the wire precision: what a first
            # multi-GPU record needs to tell a slow rank from a slow collective
            "per_rank_ms": [round(x, 4) for x in per_rank_ms],
            "rccl_ranks": rccl_ranks,
            "dp": {"buckets": eng.dp_buckets, "bucket_mb": [round(4e-6 * n, 2) for n in
                                                             ([eng.nparam - eng.enc_end, eng.enc_end - eng.enc_mid, eng.enc_mid]
                                                              if eng.dp_buckets >= 3 else [eng.nparam - eng.enc_end, eng.enc_end])],
                   "wire": "bf16" if dp.bf16_wire else "fp32"} if dp.active else None,
            "roofline": roof,
        }
        gfile = _latest_profile("gemm_roofline.txt")
        if gfile:       # per-shape GEMM roofline table of the latest round (tools/bench_gemm.py on the GPU box)
            tail = [ln.strip() for ln in open(gfile).read().splitlines() if ln.startswith(("sum:", "roofline fraction"))]
            line["gemm_roofline"] = {"source": "profiles/" + os.path.basename(gfile), "summary": tail}
        if world == 1 and not args.no_decode:
            line["decode"] = decode_bench(eng)
        if world == 1 and not args.no_cpu_baseline and args.model == "self-attention-tacotron":
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    dp.shutdown()


if __name__ == "__main__":
    main()
