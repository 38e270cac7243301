import os, sys, torch
sys.path.insert(0, "/root/repo")
import satt_amd
from satt_amd import ops
ops.set_precision("bf16")
B,T,H,hd=32,400,2,128
D=H*hd
kvq=torch.randn(B*T,3*D,device="cuda"); o=torch.empty(B*T,D,device="cuda"); lse=torch.empty(B*H,T,device="cuda")
do=torch.randn(B*T,D,device="cuda"); dkvq=torch.empty(B*T,3*D,device="cuda"); dl=torch.empty(B*H,T,device="cuda")
seed=torch.zeros(1,dtype=torch.int32,device="cuda")
drop=ops.Drop(0.05,16,seed)
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)*1e3/n
print("fwd %.1f us"%t(lambda: ops.flash_attn_fwd(kvq,D,o,lse,B,T,H,1/hd**0.5,True,drop)))
print("bwd %.1f us"%t(lambda: ops.flash_attn_bwd(kvq,D,o,do,lse,dl,dkvq,B,T,H,1/hd**0.5,True,drop)))
kb=torch.empty(B*T,3*D,dtype=torch.bfloat16,device="cuda"); dob=torch.empty(B*T,D,dtype=torch.bfloat16,device="cuda")
print("fwd + bf16 copies %.1f us"%t(lambda: ops.flash_attn_fwd(kvq,D,o,lse,B,T,H,1/hd**0.5,True,drop,kvq_b=kb)))
print("bwd, bf16 sources %.1f us"%t(lambda: ops.flash_attn_bwd(None,D,o,do,lse,dl,dkvq,B,T,H,1/hd**0.5,True,drop,kvq_b=kb,do_b=dob)))
nt=(T+63)//64
print("bwd suffix tile, bf16 sources %.1f us"%t(lambda: ops.flash_attn_bwd(None,D,o,do,lse,dl,dkvq,B,T,H,1/hd**0.5,True,drop,tiles=(nt-1,nt),kvq_b=kb,do_b=dob)))
print("bwd prefix tiles, bf16 sources %.1f us"%t(lambda: ops.flash_attn_bwd(None,D,o,do,lse,dl,dkvq,B,T,H,1/hd**0.5,True,drop,tiles=(0,nt-1),kvq_b=kb,do_b=dob)))
