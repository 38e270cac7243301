import sys, time
sys.path.insert(0, '.')
import torch
import satt_amd
from satt_amd import ops
s1 = torch.cuda.Stream()
big = torch.zeros(1 << 20, dtype=torch.int32, device="cuda")      # 4 MB block of its own
big[0] = 1000
view = big[1024:1025]                                           # offset 4096 B inside the allocation
flag = torch.zeros(1, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
ops.stream_wait_value(view, 1, s1)
with torch.cuda.stream(s1):
    flag.fill_(7)
time.sleep(0.2)
print("before release: flag =", int(flag.cpu()[0] if False else 0) if False else "(not read: would sync)")
q = s1.query()
print("s1 idle before release (True means the wait did NOT block):", q)
view.fill_(1)          # default stream
torch.cuda.synchronize()
print("after release: flag =", int(flag[0]), " s1 idle:", s1.query())
# small tensor from the caching allocator pool (what the engine uses)
small = torch.zeros(16, dtype=torch.int32, device="cuda")
flag.zero_(); torch.cuda.synchronize()
ops.stream_wait_value(small[0:1], 1, s1)
with torch.cuda.stream(s1):
    flag.fill_(9)
time.sleep(0.2)
print("small tensor: s1 idle before release:", s1.query(), " data_ptr offset in 2MB block:", small.data_ptr() % (2 << 20))
small.fill_(1); torch.cuda.synchronize()
print("after release flag =", int(flag[0]))
