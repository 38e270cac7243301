#!/usr/bin/env python
"""Event records / waits per train step and per stream (each is a marker packet: ~5 us of queue time where it sits between two
dependent kernels, tools/event_cost.py): counts them by monkey-patching torch's Event / Stream for one step."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satt_amd  # noqa: F401
from satt_amd import ops
from satt_amd.engine import Engine
from satt_amd.params import ModelConfig
from satt_amd.datasets.synthetic import synthetic_batch

ops.set_precision("bf16")
cfg = ModelConfig()
eng = Engine(cfg, "cuda", rng_seed=3)
b = eng.to_device_batch(synthetic_batch(32, 160, 800, num_mels=cfg.num_mels, r=cfg.r, seed=5))
for _ in range(3):
    eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize()
main = torch.cuda.current_stream()
rec, wait = collections.Counter(), collections.Counter()
where = collections.Counter()
orig_record, orig_wait = torch.cuda.Event.record, torch.cuda.Stream.wait_event


def record(self, stream=None):
    st = stream if stream is not None else torch.cuda.current_stream()
    rec["main" if st == main else "side"] += 1
    if st == main:
        fr = traceback.extract_stack(limit=4)[0:3]
        where["record  " + " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr))] += 1
    return orig_record(self, stream) if stream is not None else orig_record(self)


def wait_event(self, ev):
    wait["main" if self == main else "side"] += 1
    if self == main:
        fr = traceback.extract_stack(limit=4)[0:3]
        where["wait    " + " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr))] += 1
    return orig_wait(self, ev)


torch.cuda.Event.record, torch.cuda.Stream.wait_event = record, wait_event
eng.train_step(b); eng.optimizer_step()
torch.cuda.synchronize()
torch.cuda.Event.record, torch.cuda.Stream.wait_event = orig_record, orig_wait
print("event records: main %d, side %d; stream waits: main %d, side %d" % (rec["main"], rec["side"], wait["main"], wait["side"]))
for k, v in where.most_common():
    print("%3d  %s" % (v, k))
