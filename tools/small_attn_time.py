import os, sys, math, torch
sys.path.insert(0, "/root/repo")
import satt_amd
from satt_amd import ops
B,T,H,hd=32,160,2,16; D=H*hd
kvq=torch.randn(B*T,3*D,device="cuda"); o=torch.empty(B*T,D,device="cuda"); p=torch.empty(B*H,T,T,device="cuda")
do=torch.randn(B*T,D,device="cuda"); dkvq=torch.empty(B*T,3*D,device="cuda"); rs=torch.empty(B*H,T,device="cuda")
seed=torch.zeros(1,dtype=torch.int32,device="cuda"); drop=ops.Drop(0.05,16,seed)
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)*1e3/n
print("%s small attn B=32 T=160: fwd %.1f us  bwd (2 launches) %.1f us" % (os.environ.get("SATT_LIB_PATH","in-tree"),
      t(lambda: ops.small_attn_fwd(kvq,D,p,o,B,T,H,1/math.sqrt(hd),drop)), t(lambda: ops.small_attn_bwd(kvq,D,p,do,dkvq,rs,B,T,H,1/math.sqrt(hd),drop))))
