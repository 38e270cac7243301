"""Data-parallel exchange step (the reference's only parallelism: tf.contrib.distribute.MirroredStrategy behind
--multi-gpus, reference train.py:68,74).  MI355X-native form: one process per GPU, identical parameter replicas,
per-rank BatchNorm statistics, and ONE sum all-reduce of the flat fp32 gradient buffer per step, split into THREE
contiguous buckets in the order the hand-written backward finishes them - decoder parameters (10.4 MB, under the whole
encoder backward), upper encoder `enc.proj1.W .. enc.sa` (~5 MB, under the conv-bank backward), conv bank + pre-net +
embedding (9.6 MB, the exposed tail); SATT_DP_BUCKETS=2 merges the two encoder buckets - each enqueued on the
issuing stream as soon as its gradients are complete.  The optimiser
kernel divides by world size (grad_scale) and applies the global-norm clip to the averaged gradient.
xGMI is point-to-point: with a 25 MB payload the exchange is latency-dominated, so few large buckets beat many."""
import datetime
import os

import torch
import torch.distributed as dist

# c10d enqueues a BLOCKING NCCL collective on the caller's current stream only in recent releases (older ones run it on the
# process group's internal stream and make the current stream wait for it - still correct, but not the queue behaviour the
# bucket placement relies on).  Verified on 2.10 (profiles/r02_step_phases_one_rank_rccl.txt); anything older takes the
# async path, which is correct on every version.
_BLOCKING_ON_CURRENT_STREAM = tuple(int(x) for x in torch.__version__.split("+")[0].split(".")[:2]) >= (2, 10)


class DataParallel:
    def __init__(self, world=1, rank=0, local_rank=0, backend=None, grad=None, force=False, grad_dtype=None):
        """force: run the collectives even with world == 1 (a one-rank RCCL group: exercises library start-up, the
        stream hand-off of the asynchronous all-reduce and its wait on a single GPU; tests / diagnostics).
        grad_dtype="bf16" (or SATT_DP_GRAD_DTYPE=bf16): the buckets cross the links as bf16 (half the bytes: xGMI rings are
        per-link bound) - cast, SUM, cast back into the fp32 gradient; the default keeps the exchange in fp32 (bit-identical
        replicas AND a sum that does not depend on the precision of the wire)."""
        self.world, self.rank = world, rank
        self.bf16_wire = (grad_dtype or os.environ.get("SATT_DP_GRAD_DTYPE", "")).lower() in ("bf16", "bfloat16")
        self.active = world > 1 or force
        self.pending = []
        self.grad = grad
        if self.active and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
            # rank 0 runs the EVAL pass at checkpoints while the others wait at the barrier behind it (models.train): the
            # default 10-minute collective watchdog is too short for a long validation list
            kw = {"timeout": datetime.timedelta(hours=2)}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        self.on_current_stream = self.active and torch.cuda.is_available() and dist.get_backend() == "nccl" and \
            _BLOCKING_ON_CURRENT_STREAM
        self.timing = None      # set to [] to collect HIP-event brackets: ("bucket", start, end) per collective on its issuing
        #                         stream, ("wait", before, after) on the consumer stream (bench.py: allreduce_ms / exposed_allreduce_ms)

    def bind(self, grad):
        self.grad = grad
        return self

    def allreduce(self, lo, hi, grad=None):
        """SUM all-reduce of grad[lo:hi] (a contiguous bucket of the flat gradient buffer), asynchronous to the caller's stream.
        RCCL: the collective is issued as a BLOCKING op, which c10d enqueues on the CURRENT stream (the weight-gradient stream
        here) without touching the host, and an event behind it is what wait() makes the consumer wait for.  The async form
        runs on c10d's internal stream: a fifth stream on four hardware queues - measured, it shared the MAIN stream's queue,
        and its wait for the decoder's weight gradients held the encoder backward up for 0.3 ms per step."""
        if not self.active:
            return
        g = self.grad if grad is None else grad
        buf = g[lo:hi]
        if self.bf16_wire:
            buf = buf.to(torch.bfloat16)              # (on the issuing stream, in front of the collective)
        if self.on_current_stream:
            timed = self.timing is not None
            if timed:
                e0 = torch.cuda.Event(enable_timing=True); e0.record(torch.cuda.current_stream())
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=False)
            if self.bf16_wire:
                g[lo:hi].copy_(buf)
            ev = torch.cuda.Event(enable_timing=timed); ev.record(torch.cuda.current_stream())
            if timed:
                self.timing.append(("bucket", e0, ev))
            self.pending.append(ev)
        elif self.bf16_wire:
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
            self.pending.append((w, g[lo:hi], buf))
        else:
            self.pending.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))

    def wait(self):
        timed = self.timing is not None and self.pending and torch.cuda.is_available()
        if timed:       # how long the consumer stream really stalls for the exchange = the EXPOSED part of the all-reduce
            w0 = torch.cuda.Event(enable_timing=True); w0.record(torch.cuda.current_stream())
        self._wait()
        if timed:
            w1 = torch.cuda.Event(enable_timing=True); w1.record(torch.cuda.current_stream())
            self.timing.append(("wait", w0, w1))

    def timing_summary(self):
        """(total ms inside the bucket collectives on their issuing streams, total ms the consumer stream waited); call after
        torch.cuda.synchronize()"""
        t = self.timing or []
        return (sum(a.elapsed_time(b) for k, a, b in t if k == "bucket"), sum(a.elapsed_time(b) for k, a, b in t if k == "wait"))

    def _wait(self):
        for w in self.pending:
            if isinstance(w, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(w)
            elif isinstance(w, tuple):               # bf16 wire on the asynchronous path: cast back behind the wait
                w[0].wait()
                w[1].copy_(w[2])
            else:
                w.wait()
        self.pending = []

    def barrier(self):
        if self.active:
            dist.barrier()

    def max_over_ranks(self, x):
        if not self.active:
            return x
        dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def gather_over_ranks(self, x):
        """[x of rank 0, x of rank 1, ...] on every rank (bench.py: per-rank step times)"""
        if not self.active:
            return [x]
        dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        return [float(o[0]) for o in out]

    def broadcast_params(self, flat):
        """make every replica start from rank 0's parameters"""
        if self.active:
            dist.broadcast(flat, src=0)

    def shutdown(self):
        if self.active and dist.is_initialized():
            dist.destroy_process_group()
