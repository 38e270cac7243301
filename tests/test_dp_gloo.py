"""world_size-2 CPU (gloo) test of the data-parallel exchange step (parallel.py): bucketed async SUM all-reduce
of the flat gradient buffer, max-over-ranks timing reduction, parameter broadcast.  The N-GPU path uses the same
code over RCCL (backend "nccl")."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import satt_amd  # noqa: F401
    from satt_amd.parallel import DataParallel
    dp = DataParallel(world, rank, rank, backend="gloo")
    n, enc_end = 1000, 400
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    dp.bind(g)
    dp.allreduce(enc_end, n)          # decoder bucket first (ready first in the backward pass)
    dp.allreduce(0, enc_end)
    dp.wait()
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(g, expect)
    flat = torch.full((8,), float(rank))
    dp.broadcast_params(flat)
    ok = ok and bool((flat == 0).all())
    mx = dp.max_over_ranks(float(rank + 1))
    ok = ok and mx == float(world)
    ok = ok and dp.gather_over_ranks(float(rank) + 0.5) == [r + 0.5 for r in range(world)]       # bench.py's per-rank step times
    # the three-bucket order of Engine.train_step (decoder, upper encoder, conv bank + embedding), fp32 and bf16 wire
    for wire in ("fp32", "bf16"):
        dp.bf16_wire = wire == "bf16"
        g3 = (torch.arange(n, dtype=torch.float32) % 64) * (rank + 1)        # (small integers: exact in bf16)
        mid = 150
        dp.allreduce(enc_end, n, g3); dp.allreduce(mid, enc_end, g3); dp.allreduce(0, mid, g3)
        dp.wait()
        ok = ok and torch.equal(g3, (torch.arange(n, dtype=torch.float32) % 64) * sum(r + 1 for r in range(world)))
        # a poisoned bucket (satt_poison_on_error on one rank) is non-finite on every rank after the sum, whatever the wire
        gp = torch.ones(16)
        if rank == 1:
            gp[0] = float("nan")
        dp.allreduce(0, 16, gp); dp.wait()
        ok = ok and bool(torch.isnan(gp[0])) and bool(torch.isfinite(gp[1:]).all())
    dp.bf16_wire = False
    dp.barrier()
    q.put((rank, ok))
    dp.shutdown()


def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_average_of_shard_gradients_equals_full_batch_gradient():
    """The quantity the exchange step must reproduce: for a loss that is a per-sample masked mean over the
    GLOBAL batch, grad(full batch) == sum over shards of (n_shard / n_total) * grad(shard).  (BatchNorm statistics
    are per-replica in the reference's MirroredStrategy, SURVEY.md §2.2, so the check uses a BN-free loss.)"""
    g = torch.Generator().manual_seed(0)
    W = torch.randn(6, 3, generator=g, dtype=torch.float64, requires_grad=True)
    x = torch.randn(8, 6, generator=g, dtype=torch.float64); y = torch.randn(8, 3, generator=g, dtype=torch.float64)
    full = torch.autograd.grad(((x @ W - y).abs()).mean(), W)[0]
    parts = [torch.autograd.grad(((x[s] @ W - y[s]).abs()).mean(), W)[0] for s in (slice(0, 4), slice(4, 8))]
    assert torch.allclose(full, (parts[0] + parts[1]) / 2)


def _run_bench(args, env_extra=None, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)


def test_bench_gpus_n_is_one_command_and_checks_its_world_size():
    """`python bench.py --gpus 2` with WORLD_SIZE unset spawns its two ranks itself (reference train.py:16,68: --multi-gpus is one
    command) and prints `n_gpus` from the PROCESS GROUP; a world size that does not match --gpus, or too few devices, is an error
    - never a 1-GPU line labelled otherwise.  (--dry-run: the N-rank plumbing without the engine, over gloo: there is no GPU here.)"""
    import json
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["sum_ok"] and len(line["per_rank_ms"]) == 2
    assert sorted(x["rank"] for x in line["rccl_ranks"]) == [0, 1] and len({x["pid"] for x in line["rccl_ranks"]}) == 2
    # an external launcher's world size must match the flag
    r = _run_bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout) and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # without --dry-run the launcher refuses to start N ranks on fewer than N devices
    import torch
    if torch.cuda.device_count() < 2:
        r = _run_bench(["--gpus", "2"])
        assert r.returncode != 0 and "device(s) visible" in (r.stderr + r.stdout)
