import torch, math, sys
sys.path.insert(0, '.')
import satt_amd
from satt_amd import ops
from satt_amd import engine
print("same module:", engine.ops is ops, ops.__name__, engine.ops.__name__)
torch.manual_seed(0)
M,K,N=256,512,128
x=torch.randn(M,K,device='cuda'); W=torch.randn(K,N,device='cuda')/math.sqrt(K)
ref=(x.double()@W.double())
for prec in ("f32","bf16"):
    ops.set_precision(prec)
    out=torch.empty(M,N,device='cuda')
    ops.linear(x,W,None,out)
    torch.cuda.synchronize()
    print(prec, ops.get_precision(), float((out.double()-ref).abs().max()/ref.abs().max()))
