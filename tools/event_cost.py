#!/usr/bin/env python
"""GPU-side cost of an event record between two dependent kernels of one stream (the engine records one per weight-gradient
hand-off), torch events vs raw HIP events created with hipEventDisableSystemFence.  The GPU is held busy while the host enqueues
everything, so the figures are queue-processing time only."""
import ctypes
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
DISABLE_TIMING, RELEASE_TO_DEVICE, DISABLE_FENCE = 0x2, 0x40000000, 0x20000000
x = torch.zeros(1024, device="cuda"); x2 = torch.zeros(1024, device="cuda")
side = torch.cuda.Stream()
N = 400


def raw_events(flags):
    evs = []
    for _ in range(N):
        e = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(e), flags) == 0
        evs.append(e)
    return evs


def run(mode, evs=None):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream().cuda_stream
    torch.cuda._sleep(int(2e8))
    a.record()
    for i in range(N):
        x.add_(1.0)
        if mode == 1:
            ev = torch.cuda.Event(); ev.record()
        elif mode == 2:
            ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        elif mode == 3:
            hip.hipEventRecord(evs[i], main)
        elif mode == 4:
            hip.hipEventRecord(evs[i], main); hip.hipStreamWaitEvent(side.cuda_stream, evs[i], 0)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / N


print("%-52s %.2f us per kernel" % ("kernels only", (run(0), run(0))[1]))
print("%-52s %.2f us" % ("+ torch event record", (run(1), run(1))[1]))
print("%-52s %.2f us" % ("+ torch event record + side stream wait", (run(2), run(2))[1]))
done = torch.cuda.Event(); done.record(side); torch.cuda.synchronize()


def run_wait():
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(2e8))
    a.record()
    for i in range(N):
        x.add_(1.0)
        torch.cuda.current_stream().wait_event(done)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / N


print("%-52s %.2f us" % ("+ main stream waits on an already complete event", (run_wait(), run_wait())[1]))
for name, fl in (("disable timing", DISABLE_TIMING), ("disable timing | disable system fence", DISABLE_TIMING | DISABLE_FENCE),
                 ("disable timing | release to device", DISABLE_TIMING | RELEASE_TO_DEVICE)):
    evs = raw_events(fl)
    print("%-52s %.2f us" % ("+ raw event (%s)" % name, (run(3, evs), run(3, evs))[1]))
    print("%-52s %.2f us" % ("+ raw event (%s) + side wait" % name, (run(4, evs), run(4, evs))[1]))
