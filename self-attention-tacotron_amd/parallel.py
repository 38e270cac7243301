"""Data-parallel exchange step (the reference's only parallelism: tf.contrib.distribute.MirroredStrategy behind
--multi-gpus, reference train.py:68,74).  MI355X-native form: one process per GPU, identical parameter replicas,
per-rank BatchNorm statistics, and ONE sum all-reduce of the flat fp32 gradient buffer per step, split into two
contiguous buckets (decoder parameters, then encoder parameters) that are launched on RCCL's stream as soon as
the hand-written backward has finished them, so the decoder bucket overlaps the encoder backward.  The optimiser
kernel divides by world size (grad_scale) and applies the global-norm clip to the averaged gradient.
xGMI is point-to-point: with a 25 MB payload the exchange is latency-dominated, so few large buckets beat many."""
import os

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, world=1, rank=0, local_rank=0, backend=None, grad=None, force=False):
        """force: run the collectives even with world == 1 (a one-rank RCCL group: exercises library start-up, the
        stream hand-off of the asynchronous all-reduce and its wait on a single GPU; tests / diagnostics)"""
        self.world, self.rank = world, rank
        self.active = world > 1 or force
        self.pending = []
        self.grad = grad
        if self.active and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    def bind(self, grad):
        self.grad = grad
        return self

    def allreduce(self, lo, hi, grad=None):
        """async SUM all-reduce of grad[lo:hi] (a contiguous bucket of the flat gradient buffer)."""
        if not self.active:
            return
        g = self.grad if grad is None else grad
        self.pending.append(dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def wait(self):
        for w in self.pending:
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

    def broadcast_params(self, flat):
        """make every replica start from rank 0's parameters"""
        if self.active:
            dist.broadcast(flat, src=0)

    def shutdown(self):
        if self.active and dist.is_initialized():
            dist.destroy_process_group()
